// Types of the device-resident HET cache shared by its translation units: csrc/cache.hip (the call-by-call flows) and
// csrc/cache_block.hip (the planned flow: the bookkeeping of a block of batches runs ahead on a side stream, every lookup and
// every update is ONE launch).  Reference: src/hetu_cache; see the header of cache.hip.
#pragma once
#include "plan_dev.h"

#include <utility>
#include <vector>

namespace ha {

enum LineState : uint8_t { kFree = 0, kResident = 1, kEvictedDirty = 2, kTransient = 3, kPending = 4,
                           kStored = 5 /* LFUOpt permanent store */ };
enum Policy { kLRU = 0, kLFU = 1, kLFUOpt = 2 };
constexpr int kUseCntMax = 10;  // lfuopt_cache.h:26

struct CacheCtl {
    long long size;       // resident lines
    long long free_top;   // entries in free_list
    long long log_head, log_tail;
    long long evict_n;    // lines waiting in the evict list
    long long clock;      // next stamp
    // per-call scratch
    long long U, M, nhit, E, pulled, C, dropped;
    long long n_base;      // LFU/LFUOpt: resident lines in the lowest use bucket
    long long n_hash;      // LFUOpt: resident lines outside the permanent store
    long long parked[4];   // push_pull: U, M, nhit of the parked pull phase; U of the push phase
    long long scan_victim; // slot of the lowest (use, stamp) line outside the lowest bucket, or -1
    // last op report: type(0 pull,1 push), num_all, num_unique, num_miss, num_transfered, num_evict, is_full
    long long perf[8];     // [7]: pushed lines of a cache_update_same_post_kernel update, to be added to [4]
    long long out_n;       // remote mode: outbox entries of the last update (U + E), -1 on overflow
    unsigned long long ph[16];   // ha_cache_phase_times: 100 MHz clock at the phase boundaries of the last lookup's bookkeeping
    unsigned long long fb_xw[64];     // cache_finish_book_kernel, per finish chunk: valid << 63 | heads << 32 | pulls << 16 | misses
    long long fb_timeout;             // sticky: a workgroup of cache_finish_book_kernel gave up waiting for another one's word
    long long snap[4];     // {clock, log_tail, free_top, evict_n} as the last lookup left them (cache_update_same_post_kernel)
    long long book_seq;    // cache_book_block_kernel (cache_block.hip): number of the last exchange between its workgroups
};

// Per-line bookkeeping as ONE 32-byte record: the bookkeeping kernels reach lines at random slots from a single
// workgroup, where every separate array costs its own address translation per line (the eviction walk read stamp,
// state, updates and key of ~1,000 victims from four arrays: 10 us of one compute unit's time at the criteo batch).
struct alignas(32) LineMeta {
    unsigned long long stamp;
    long long version;
    uint32_t key;
    int32_t updates;
    int32_t freq;
    uint8_t state;
    uint8_t hg;        // the bookkeeping's copy of hasgrad[slot]: the planned flow (cache_block.hip) decides from it a block ahead of the rows
    uint8_t pad[2];
};
static_assert(sizeof(LineMeta) == 32, "one line record = 32 bytes");

struct Cache {
    int policy;
    int64_t limit, length, width, nmax, S, Lcap;
    int64_t pull_bound, push_bound;
    bool bypass;
    CacheCtl *ctl;
    int32_t *slot_of;
    LineMeta *line;        // [S] key / version / updates / freq / state / stamp of the line in slot s
    uint8_t *hasgrad;      // [S] dense: the accumulate kernels read it through their own row map (ApplyMaps::dst_init)
    float *data, *grad;
    int32_t *free_list;
    uint32_t *log_slot;
    unsigned long long *log_stamp;
    int32_t *evict_slots;
    // per-call scratch, sized nmax
    void *plan_ws, *plan2_ws;
    int32_t *uslot, *data_row;
    uint32_t *flag, *rank;
    uint8_t *pushflag;
    uint32_t *pushkeys_u32;
    // second scratch set (the pull phase of push_pull)
    void *plan_ws_b;
    int32_t *uslot_b, *data_row_b;
    uint32_t *flag_b, *rank_b;
    uint8_t *pushflag_b;
    // LFU / LFUOpt on a large cache: the argmin over the resident lines by kScanParts workgroups (cache_scan_victim_part_kernel),
    // a partial result each; the single-workgroup bookkeeping then reduces these instead of walking every line
    unsigned long long *scan_key;
    int32_t *scan_slot;
    // store (the "server"): rows [row_start, row_start + store_rows) of the global table
    float *table;
    long long *srv_ver;
    int64_t store_rows, row_start;
    // REMOTE store (rows owned by other ranks, or kept in host memory): the cache never touches the store
    // itself.  A lookup exports (key, cached version) of its unique keys, the store's owner takes
    // syncEmbedding's decision and the answer arrives in the INBOX; an update leaves the lines to push
    // (pushEmbedding) in the OUTBOX.  herald_amd/cache.py moves both (sharded.py exchange / host staging).
    int remote;
    uint32_t *req_keys;    // [nmax] unique keys of the batch
    long long *req_ver;    // [nmax] cached version of each (-1: no data yet)
    int32_t *inbox_pull;   // [nmax] the owner's decision (1 = row follows)
    int32_t *inbox_idx;    // [nmax] its row in inbox_rows
    long long *inbox_ver;  // [nmax] server version of the row
    float *inbox_rows;     // [nmax, width]
    int64_t out_cap;       // outbox entries: batch lines at [0, U), pending evicted lines at [U, U + E)
    uint32_t *out_keys;    // kNoPush = entry not pushed
    int32_t *out_upd;
    float *out_rows;       // [out_cap, width]
};
constexpr uint32_t kNoPush = 0xFFFFFFFFu;
template <typename T>
static int dmalloc(T **p, size_t count) {
    HA_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T) + 256));
    return 0;
}
// Zeroes of a fresh allocation, COMPLETE when the call returns.  hipMemset on device memory is a launch on the null stream
// that the host does not wait for, and the streams this library runs on are non-blocking ones (no implicit order with the null
// stream): without the wait the zeroes can land after the first kernel that writes the buffer (a plan header's n_unique
// went back to 0 that way, one new cache in eight).
static int dzero(void *p, size_t bytes) {
    HA_CHECK_HIP(hipMemsetAsync(p, 0, bytes, nullptr));
    HA_CHECK_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

// ---- the planned flow (cache_block.hip) ----
constexpr int kPlanBlockMax = 16;     // batches per planned block
// what the bookkeeping launch leaves per planned batch (the perf dict's counts; pulled / npush are counted on demand)
struct PlanRec {
    long long n, U, M, E, evicted, size, full, npush, pulled;
    // the LFU policies (LRU: erep = E, umiss = 0, vh_slot = -1): erep = dirty lines the lookup evicted (the perf dict's
    // num_evict; E counts those of them the update pushes by a wave of their own), umiss = keys the update does not find in
    // the cache (num_miss of the Push record), vh_* = the evicted line when it is a line of the batch itself (slot, update
    // counter; its key's record carries kPosVictim)
    long long erep, umiss, vh_slot, vh_upd;
};
// the buffers of one planned block (two exist: the block being consumed and the next)
struct PlanSlot {
    void *ws[kPlanBlockMax] = {};       // index plans of the block's batches
    int32_t *it_slot = nullptr;         // [kPlanBlockMax][nmax] items per unique key, see BookArgs (cache_block.hip)
    uint8_t *it_flag = nullptr;
    int32_t *it_upd = nullptr;
    int32_t *it_upd_pos = nullptr;      // [kPlanBlockMax][nmax] it_upd per sorted position
    int4 *pos_item = nullptr;           // [kPlanBlockMax][nmax] the same per SORTED POSITION: {slot, key, kPos* flags, occurrence index}
    long long *pver = nullptr;          // [kPlanBlockMax][nmax] line versions staged by the lookup for the update
    int32_t *ev_slot = nullptr;         // [kPlanBlockMax][nmax] evicted dirty lines per batch
    uint32_t *ev_key = nullptr;
    int32_t *ev_upd = nullptr;
    PlanRec *rec = nullptr;             // [kPlanBlockMax]
    int64_t n[kPlanBlockMax] = {};
    int count = 0;                      // batches of the block (0: the slot was never used)
    int next_call = 0;                  // 2 i = the lookup of batch i is due, 2 i + 1 = its update; 2 count = consumed
    bool waited = false;                // the row stream already waits for `booked`
    hipEvent_t booked = nullptr;        // behind the bookkeeping launch
    hipStream_t booked_on = nullptr;
    hipEvent_t rows_done = nullptr;     // behind the block's last row launch (recorded by its last ha_cache_update_planned)
    bool rows_recorded = false;
};
}  // namespace ha

struct ha_cache;
namespace ha {
int cache_perf_planned(ha_cache *h, int64_t *out_host, hipStream_t s);
}

struct ha_cache {
    ha::Cache c;
    std::vector<void *> allocs;
    int64_t plan_n = -1;   // n of the lookup whose plan is still in plan_ws (ha_cache_update_same_keys)
    int64_t pp_pull = -1, pp_push = 0;   // sizes of the push_pull between its begin and finish (remote store)
    int64_t out_pad = 0;   // remote store: the updates mark outbox entries [U + E, out_pad) as not pushed
    // cache_update_same_post_kernel's preconditions, tracked on the host (no read-back): evict_empty = the last call
    // was an update (it pushes every pending evicted line); same_fast = the plan in plan_ws belongs to a lookup that
    // started from an empty evict list on an LRU cache with limit >= max_batch and a local store
    bool evict_empty = true, same_fast = false;
    int fused_update = 7;  // HA_CACHE_FUSED: bit 0 = the two-launch update, bit 1 = eviction beside the lookup's row copies,
                           // bit 2 = the plan's finish does the lookup's bookkeeping (cache_finish_book_kernel)
    int64_t fused_count = 0;
    // stage times of the last call (ha_cache_set_timing): HIP events between the launches of a call -- the GPU analogue of
    // the reference's std::chrono stamps between the stages of _embeddingLookup / _embeddingUpdate (cache.cc:61-106,133-196)
    bool timing = false;
    hipEvent_t tev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    unsigned tmask = 0;
    // ha_cache_sort_ahead: the stable sort of the NEXT lookup's keys runs on a stream of the cache's own, into a second plan
    // workspace, beside the calls of the current batch (the sort reads the keys only -- the reference's data loader hands the
    // ids over a batch early as well, dataloader.py:63-98)
    void *plan_ws_alt = nullptr;
    hipStream_t ahead_stream = nullptr;
    hipEvent_t ahead_fork = nullptr, ahead_join = nullptr;
    const void *ahead_keys = nullptr;
    int64_t ahead_n = -1;
    int ahead_kind = -1;
    // ha_cache_sort_ahead_batch: the sorts of the next kAheadRing lookups in ONE launch on the caller's stream, into a ring
    // of plan workspaces the lookups take in order
    static constexpr int kAheadRing = 16;
    size_t plan_bytes = 0;
    void *ring_ws[kAheadRing] = {};
    const void *ring_keys[kAheadRing] = {};
    int64_t ring_n[kAheadRing] = {};
    int ring_kind = -1, ring_head = 0, ring_count = 0;
    // the planned flow (cache_block.hip)
    // THREE slots for at most two outstanding blocks: the bookkeeping of block b + 1 reuses the buffers of block b - 2, whose
    // rows are long done -- its wait for them (an event) passes at once.  With two slots it waited for the rows of block b - 1,
    // i.e. for most of a block's duration, as a barrier parked at the head of the (high-priority) planning stream's hardware
    // queue -- and such a barrier makes the row stream's launches take two to three times as long when the two streams'
    // hardware queues are an unlucky pair (which they are for one in four orders of stream creation: profiles/r06/
    // cache_tier_third_instance.txt, docs/EXPERIMENTS.md round 6 section 12).
    static constexpr int kPlanSlots = 3;
    ha::PlanSlot plan[kPlanSlots];
    int plan_next = 0;                  // blocks planned so far (slot = plan_next % kPlanSlots)
    unsigned long long *plan_xw = nullptr;   // exchange words of the bookkeeping launch
    hipEvent_t plan_fork = nullptr;
    // LFU / LFUOpt planned: the victim tree (cache_block.hip) -- the (use, stamp) key of every slot and the minimum of every
    // block of 32 slots -- valid while only planned calls touch the cache (any call-by-call entry point clears lfu_tree_ok)
    unsigned long long *lfu_lkey = nullptr, *lfu_bmin = nullptr, *lfu_xk = nullptr;
    long long *lfu_xb = nullptr;
    long long lfu_nblk = 0;
    bool lfu_tree_ok = false;
    ha::PlanSlot *last_planned = nullptr;    // the last planned call: slot, batch, type (0 lookup / 1 update), for ha_cache_perf
    int last_planned_idx = 0, last_planned_type = -1;
};

enum { kTStart = 0, kTSort = 1, kTLookup = 2, kTCopy = 3, kTTransfer = 4, kTEnd = 5 };
static inline void cache_mark(ha_cache *h, int slot, hipStream_t s, bool first = false) {
    if (!h->timing)
        return;
    if (first)
        h->tmask = 0;
    if (hipEventRecord(h->tev[slot], s) == hipSuccess)
        h->tmask |= 1u << slot;
}
extern "C" int ha_cache_plan_pending(ha_cache *h);
