"""Synthetic Criteo-shaped inputs for tests and bench (host-side numpy; deterministic by seed).

The reference trains on Criteo with ONE global table of 33,762,577 rows: 26 categorical fields,
label-encoded and offset into one id space (examples/ctr/models/load_data.py:193-205;
examples/ctr/models/wdl_criteo.py:9).  No dataset travels with this repo, so ids are drawn per
field from a truncated zipf over that field's cardinality.  The 26 cardinalities are the public
Criteo-Kaggle label-encoding counts; they sum to exactly 33,762,577.  The exponent is calibrated
once so that a bs=256 batch has about 41 % unique ids, the statistic the reference publishes
(examples/ctr/torch_models/README.md:15,17: 2736 unique of 6656).
"""
import numpy as np

CRITEO_ROWS = 33762577
CRITEO_FIELDS = np.array([
    1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992,
    5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572], dtype=np.int64)
assert int(CRITEO_FIELDS.sum()) == CRITEO_ROWS
ZIPF_ALPHA = 1.11  # calibrated: U/N = 0.41 at bs=256 (reference 0.411), 0.35 at bs=512 (0.362)


def field_layout(rows=CRITEO_ROWS, nfields=26):
    """(cardinality[f], offset[f]) scaled so that sum(card) == rows."""
    if rows == CRITEO_ROWS and nfields == 26:
        card = CRITEO_FIELDS.copy()
    else:
        base = CRITEO_FIELDS[np.arange(nfields) % 26].astype(np.float64)
        card = np.maximum(1, np.floor(base * (rows / base.sum()))).astype(np.int64)
        # hand the rounding remainder to the largest field; keep every field >= 1
        card[np.argmax(card)] += rows - int(card.sum())
        assert card.min() >= 1, "rows too small for %d fields" % nfields
    off = np.concatenate([[0], np.cumsum(card)[:-1]])
    return card, off


def _zipf_in_range(rng, card, size, alpha):
    """Truncated zipf rank in [0, card) by inverse-CDF of the continuous bounded power law."""
    u = rng.random(size)
    if abs(alpha - 1.0) < 1e-9:
        x = np.exp(u * np.log(card + 1.0))
    else:
        a = 1.0 - alpha
        x = ((card + 1.0) ** a * u + (1.0 - u)) ** (1.0 / a)
    r = np.floor(x).astype(np.int64) - 1
    return np.clip(r, 0, card - 1)


def criteo_batch(batch_size, step=0, rows=CRITEO_ROWS, nfields=26, alpha=ZIPF_ALPHA, seed=123):
    """[batch_size, nfields] int64 global row ids of one mini-batch."""
    card, off = field_layout(rows, nfields)
    rng = np.random.default_rng(seed + step)
    ids = np.empty((batch_size, nfields), dtype=np.int64)
    for f in range(nfields):
        rank = _zipf_in_range(rng, int(card[f]), batch_size, alpha)
        # fixed per-field multiplicative hash so hot ranks are spread over the field's range
        mult = 2654435761 % int(card[f]) if card[f] > 1 else 0
        if card[f] > 1 and np.gcd(mult, int(card[f])) != 1:
            mult = 1
        ids[:, f] = off[f] + (rank * max(mult, 1)) % card[f]
    return ids


def as_f32_ids(ids):
    """The operator boundary hands ids over as float32 (python/hetu/dataloader.py:14); ids above
    2^24 are rounded by this cast exactly as numpy rounds them in the reference."""
    return ids.astype(np.float32)


def grads(n, width, step=0, seed=456):
    rng = np.random.default_rng(seed + step)
    return rng.standard_normal((n, width), dtype=np.float32)


def table(rows, width, seed=123, stddev=0.01):
    """init.random_normal(stddev=0.01) (examples/ctr/models/wdl_criteo.py:13)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((rows, width), dtype=np.float32) * np.float32(stddev))
