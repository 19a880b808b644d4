/*
 * oracle.c -- CPU restatement of the reference's embedding hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under herald_amd/ may import, link or call this file; it is
 * the checker that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg compare the
 * HIP path against (and time beside it).  Each function cites the reference code it follows
 * (paths relative to /root/reference).
 *
 * Parity pin: PINNED to the reference compiled here.  oracle/build_ref.sh compiles the reference's own
 * src/dnnl_ops/EmbeddingLookup.cpp and src/dnnl_ops/Optimizers.cpp unchanged (oneDNN's real
 * <dnnl.hpp> ships with PyTorch: torch/include/dnnl.hpp) into oracle/_ref/libref_dnnl.so;
 * tests/golden/make_golden.py runs cpu_EmbeddingLookup / cpu_SGDOptimizerSparseUpdate from it and
 * commits tests/golden/dnnl_ops.json; tests/test_golden.py holds oracle_embedding_lookup and
 * oracle_sgd_sparse_update to those vectors bit for bit (and live against the .so where it exists).
 * The dedup order is pinned the same way by oracle/_ref's hetu::Unique<T> (tests/golden/unique.json).
 *
 * Build: gcc -O3 -fopenmp -ffp-contract=off -fPIC -shared oracle.c  (see herald_amd/_build.py).
 * The reference is built with "-O3 -Wall" for baseline x86-64 (CMakeLists.txt:15), i.e. without
 * FMA contraction; -ffp-contract=off states that explicitly.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

/* src/dnnl_ops/EmbeddingLookup.cpp:16-35 -- OpenMP loop of memcpy(row) per id,
 * row index = size_t(index[i]). */
int oracle_embedding_lookup(const float *table, size_t width, const float *ids,
                            size_t n, float *out) {
    const size_t entry = width * sizeof(float);
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i)
        memcpy(out + i * width, table + (size_t)ids[i] * width, entry);
    return 0;
}

/* src/dnnl_ops/Optimizers.cpp:51-74 -- deliberately serial ("to avoid lock, here not use
 * parallel"), occurrence order, param -= lr * value with two roundings. */
int oracle_sgd_sparse_update(float *table, size_t width, const float *ids,
                             size_t n, const float *grads, float lr) {
    for (size_t i = 0; i < n; ++i) {
        const size_t dst = (size_t)ids[i] * width;
        const size_t src = i * width;
        for (size_t j = 0; j < width; ++j)
            table[dst + j] -= lr * grads[src + j];
    }
    return 0;
}

/* ---- sorted unique + inverse ------------------------------------------------
 * src/hetu_cache/include/unqiue_tools.h:9-48 (argsort, then walk: push a key when it differs
 * from its sorted predecessor, map[args[i]] = size-1) == np.unique(return_inverse=True)
 * (python/hetu/ndarray.py:534,559) == the std::map of PSAgent.h:133-137. */
typedef struct {
    uint64_t key;
    size_t pos;
} kp_t;

static int kp_cmp(const void *a, const void *b) {
    const kp_t *x = (const kp_t *)a, *y = (const kp_t *)b;
    if (x->key != y->key)
        return x->key < y->key ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

/* keys[n] -> uniq[<=n] ascending, inverse[n], counts[<=n]; returns U. */
size_t oracle_unique_u64(const uint64_t *keys, size_t n, uint64_t *uniq,
                         int64_t *inverse, int64_t *counts) {
    if (n == 0)
        return 0;
    kp_t *a = (kp_t *)malloc(n * sizeof(kp_t));
    for (size_t i = 0; i < n; ++i) {
        a[i].key = keys[i];
        a[i].pos = i;
    }
    qsort(a, n, sizeof(kp_t), kp_cmp);
    size_t u = 0;
    for (size_t i = 0; i < n; ++i) {
        if (i == 0 || a[i].key != a[i - 1].key) {
            uniq[u] = a[i].key;
            if (counts)
                counts[u] = 0;
            ++u;
        }
        inverse[a[i].pos] = (int64_t)(u - 1);
        if (counts)
            counts[u - 1] += 1;
    }
    free(a);
    return u;
}

/* float32 ids -> integer keys exactly as every reference entry point does:
 * (size_t)ids[i] (EmbeddingLookup.cpp:31), (cache_key_t)keys[i] (cache.cc:51-54). */
void oracle_ids_to_keys(const float *ids, size_t n, uint64_t *keys) {
    for (size_t i = 0; i < n; ++i)
        keys[i] = (uint64_t)ids[i];
}

/* python/hetu/ndarray.py:556-576 (IndexedSlices.cpu_deduplicate): new_values zero-filled, then
 * `for i, ind in enumerate(inverse): new_values[ind] += flatten[i]` -- occurrence order.  The same
 * order is used by PSAgent::vecPushSparse (PSAgent.h:146-160, cp_val zero-initialised). */
int oracle_dedup_reduce(const int64_t *inverse, size_t n, const float *grads,
                        size_t width, size_t n_unique, float *reduced) {
    memset(reduced, 0, n_unique * width * sizeof(float));
    for (size_t i = 0; i < n; ++i) {
        float *dst = reduced + (size_t)inverse[i] * width;
        const float *src = grads + i * width;
        for (size_t j = 0; j < width; ++j)
            dst[j] += src[j];
    }
    return 0;
}

/* Server side of a sparse push: PSHandler::serve(SparsePush), ps-lite/include/ps/server/
 * PSFHandle.h:130-164 -- value[offset[j]*width + k] += vals[j*width + k] for unique offsets. */
int oracle_push_apply(float *table, size_t width, const uint64_t *uniq,
                      size_t n_unique, const float *reduced) {
    for (size_t j = 0; j < n_unique; ++j) {
        float *dst = table + (size_t)uniq[j] * width;
        const float *src = reduced + j * width;
        for (size_t k = 0; k < width; ++k)
            dst[k] += src[k];
    }
    return 0;
}

/* ps-lite/include/ps/partitioner.h:46-57 (AveragePartitioner::partitionDense): shard i holds
 * length/S + (i < length%S) contiguous rows.  starts[S+1]. */
void oracle_partition(size_t length, size_t nshard, size_t *starts) {
    const size_t per = length / nshard, rem = length % nshard;
    size_t cur = 0;
    for (size_t i = 0; i < nshard; ++i) {
        starts[i] = cur;
        cur += per + (i < rem);
    }
    starts[nshard] = cur;
}

int oracle_num_threads(void) {
    return omp_get_max_threads();
}
