// mpcombi_hip.hip -- C ABI (include/mpcombi.h) over the gfx950 kernels in kernels.hpp.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared mpcombi_hip.hip -o libmpcombi_hip.so
//
// Host side of the hot path: owns the device-resident program blocks, the frontier (to_check), the pruned list
// (murder_list) and the per-level result buffers; launches the kernels on one HIP stream and times them with HIP
// events.  No CPU fallback exists: without a usable HIP device every entry point fails with MPC_ERR_HIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <unordered_map>
#include <map>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <functional>
#include <chrono>
#include <atomic>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/mpcombi.h"
#include "kernels.hpp"
#include "kernels2.hpp"
#include "locate.hpp"
#include "graph.hpp"
#include "qp.hpp"
#include <memory>

#include "batch_level.hpp"
#include "setup_mfma.hpp"
#include <rocprim/device/device_radix_sort.hpp>

using namespace mpc;

namespace {

thread_local std::string g_error;

// ---- process-wide recycling of device blocks, pinned host blocks, streams and events -----------------------------------
// A caller that solves many small programs one after the other (the mixed-integer enumeration: one mpc_create /
// mpc_destroy per binary fixation) would otherwise pay ~50 hipMalloc/hipFree, 8 hipHostMalloc/hipHostFree and two
// stream creations per program -- 12 ms, several times the kernel time of such a program.  Blocks are kept in size
// classes (powers of two up to 1 MiB, MiB multiples above) and handed to the next handle on the same device; a block
// is only returned to the pool once the work that used it has completed (destroy synchronises its streams first).
std::mutex g_dev_pool_mutex;
std::multimap<std::pair<int, size_t>, void *> g_dev_pool_free;   // (device, size) -> block
size_t g_dev_pool_bytes = 0;
// Free blocks kept: MPC_DEV_POOL_GB (default 48 -- a sixth of the 288 GB of an MI355X; a batch of 64 sub-programs of the
// mixed-integer enumeration holds ~20 GB of level buffers at once, and the next enumeration takes them over as they are).
const size_t DEV_POOL_MAX_BYTES = [] { const char *ev = std::getenv("MPC_DEV_POOL_GB"); const double gb = ev ? std::atof(ev) : 48.0; return (size_t)(std::max(gb, 0.0) * 1073741824.0); }();
constexpr size_t DEV_POOL_MAX_BLOCK = size_t(8) << 30;

// size classes: powers of two up to 1 MiB, above that four steps per octave (1, 1.25, 1.5, 1.75 x 2^k: at most 25 % slack, and
// buffers that grow with the level -- a different size at every level of every program -- meet blocks of earlier handles)
size_t dev_size_class(size_t bytes) {
    if (bytes <= 256) return 256;
    if (bytes <= (size_t(1) << 20)) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
    size_t p = size_t(1) << 20;
    while ((p << 1) <= bytes) p <<= 1;
    const size_t step = p >> 2;
    return (bytes + step - 1) / step * step;
}
hipError_t dev_pool_take(size_t cls, void **out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_dev_pool_mutex);
        auto it = g_dev_pool_free.find({dev, cls});
        if (it != g_dev_pool_free.end()) { *out = it->second; g_dev_pool_bytes -= cls; g_dev_pool_free.erase(it); return hipSuccess; }
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {   // out of memory: give the pool back to the driver and try once more
        std::vector<void *> drop;
        { std::lock_guard<std::mutex> lk(g_dev_pool_mutex); for (auto &kv : g_dev_pool_free) if (kv.first.first == dev) drop.push_back(kv.second);
          for (auto it = g_dev_pool_free.begin(); it != g_dev_pool_free.end();) { if (it->first.first == dev) { g_dev_pool_bytes -= it->first.second; it = g_dev_pool_free.erase(it); } else ++it; } }
        for (void *q : drop) (void)hipFree(q);
        (void)hipGetLastError();
        e = hipMalloc(out, cls);
    }
    return e;
}
void dev_pool_give(void *p, size_t cls) {
    if (!p) return;
    if (cls <= DEV_POOL_MAX_BLOCK && cls == dev_size_class(cls)) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(g_dev_pool_mutex);
        if (g_dev_pool_bytes + cls <= DEV_POOL_MAX_BYTES) { g_dev_pool_free.emplace(std::make_pair(dev, cls), p); g_dev_pool_bytes += cls; return; }
    }
    (void)hipFree(p);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // != nullptr: a block this buffer has outgrown is parked there instead of being waited for -- the owner (a handle) gives the parked
    // blocks back to the pool behind its next synchronisation of all its streams (graveyard_flush).  A handle's first solve grows some
    // twenty buffers on every level: one stream synchronisation each (round 4: 170 growths per first solve of config 4).
    std::vector<std::pair<void *, size_t>> *parked = nullptr;
    hipError_t ensure(size_t bytes, hipStream_t st, bool keep = false) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = dev_size_class(std::max(bytes, cap + cap / 2));
        void *q = nullptr;
        hipError_t e = dev_pool_take(want, &q);
        if (e != hipSuccess) return e;
        if (p) {
            if (keep && cap) {
                e = hipMemcpyAsync(q, p, cap, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
            // the old block may still be read by work queued on this stream: it goes back to the pool only afterwards
            if (parked) parked->emplace_back(p, cap);
            else {
                e = st ? hipStreamSynchronize(st) : hipDeviceSynchronize();
                if (e != hipSuccess) return e;
                dev_pool_give(p, cap);
            }
        }
        p = q;
        cap = want;
        return hipSuccess;
    }
    // the caller has synchronised whatever used the block
    void release() { if (p) dev_pool_give(p, cap); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// pooled page-locked host memory (also behind mpc_host_alloc / mpc_host_free)
// Two classes of blocks that are never mixed: class 0 = staging for DMA copies (default, coarse-grained mapping); class 1 =
// blocks a RUNNING kernel writes and the host polls (streamed region records, chunk flags, the list-length counters):
// those are allocated hipHostMallocCoherent | hipHostMallocMapped, i.e. fine-grained, so that visibility of the kernel's
// system-scope release does not depend on HIP_HOST_COHERENT or on an undocumented L2 write-back.
std::mutex g_pool_mutex;
std::multimap<size_t, void *> g_pool_free[2];       // [class] size -> block
std::unordered_map<void *, std::pair<size_t, int>> g_pool_live;     // block -> (size, class)
size_t g_pool_free_bytes = 0;
constexpr size_t POOL_MAX_FREE = size_t(16) << 30;

hipError_t host_pool_take(size_t bytes, void **out, size_t *got, bool coherent = false) {
    static const bool coarse_only = [] { const char *ev = std::getenv("MPC_HOST_COARSE"); return ev && ev[0] == '1'; }();   // A/B switch
    if (coarse_only) coherent = false;
    const size_t gran = bytes >= (size_t(1) << 20) ? (size_t(1) << 20) : (size_t(64) << 10);
    const size_t need = std::max<size_t>((bytes + gran - 1) / gran * gran, gran);
    const int cls = coherent ? 1 : 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        auto it = g_pool_free[cls].lower_bound(need);
        if (it != g_pool_free[cls].end() && it->first <= need + need / 2) {
            *out = it->second;
            if (got) *got = it->first;
            g_pool_live[it->second] = {it->first, cls};
            g_pool_free_bytes -= it->first;
            g_pool_free[cls].erase(it);
            return hipSuccess;
        }
    }
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, need, coherent ? (hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocPortable);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    g_pool_live[p] = {need, cls};
    *out = p;
    if (got) *got = need;
    return hipSuccess;
}
bool host_pool_give(void *p) {
    size_t sz = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        auto it = g_pool_live.find(p);
        if (it == g_pool_live.end()) return false;
        sz = it->second.first;
        const int cls = it->second.second;
        g_pool_live.erase(it);
        if (g_pool_free_bytes + sz <= POOL_MAX_FREE) {
            g_pool_free[cls].emplace(sz, p);
            g_pool_free_bytes += sz;
            return true;
        }
    }
    (void)hipHostFree(p);
    return true;
}

// pinned (page-locked) host staging: device-to-host copies run at link speed and never page-fault
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool coherent = false;   // class 1 of the pool: written by running kernels, polled by the host
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        const size_t want = std::max(bytes, cap + cap / 2);
        release();
        return host_pool_take(want, &p, &cap, coherent);
    }
    void release() { if (p) (void)host_pool_give(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// streams and events
std::mutex g_sync_pool_mutex;
std::map<int, std::vector<hipStream_t>> g_stream_pool;
std::map<std::pair<int, int>, std::vector<hipEvent_t>> g_event_pool;   // (device, timing?) -> events
std::map<int, int> g_cu_count;

hipError_t pooled_stream(hipStream_t *out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_sync_pool_mutex);
        auto &v = g_stream_pool[dev];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void return_stream(hipStream_t s) {   // synchronised by the caller
    if (!s) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_sync_pool_mutex);
    auto &v = g_stream_pool[dev];
    if (v.size() < 1024) v.push_back(s); else (void)hipStreamDestroy(s);   // a stream costs ~3 ms to create and ~2 ms to destroy; a batch of 64 programs holds 200
}
hipError_t pooled_event(hipEvent_t *out, bool timing) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_sync_pool_mutex);
        auto &v = g_event_pool[{dev, timing ? 1 : 0}];
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    return timing ? hipEventCreate(out) : hipEventCreateWithFlags(out, hipEventDisableTiming);
}
void return_event(hipEvent_t e, bool timing) {
    if (!e) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_sync_pool_mutex);
    auto &v = g_event_pool[{dev, timing ? 1 : 0}];
    if (v.size() < 8192) v.push_back(e); else (void)hipEventDestroy(e);
}
// hipGetDeviceCount costs ~80 us per call on this runtime (it is asked at every mpc_create and every LP batch)
int device_count_cached() {
    static const int n = [] { int v = 0; return hipGetDeviceCount(&v) == hipSuccess ? v : 0; }();
    if (n > 0) return n;
    int v = 0;      // none seen yet: ask again (a device may have been made visible since)
    return hipGetDeviceCount(&v) == hipSuccess ? v : 0;
}
int cu_count(int device) {
    {
        std::lock_guard<std::mutex> lk(g_sync_pool_mutex);
        auto it = g_cu_count.find(device);
        if (it != g_cu_count.end()) return it->second;
    }
    hipDeviceProp_t prop;
    int n = 256;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
    std::lock_guard<std::mutex> lk(g_sync_pool_mutex);
    g_cu_count[device] = n;
    return n;
}

int odd_at_least(int v) { return (v % 2) ? v : v + 1; }

}  // namespace

struct mpc_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // side stream: retry kernels of few long-running wavefronts overlap the main pipeline
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_xfork = nullptr, ev_xjoin = nullptr;
    hipStream_t stream3 = nullptr;   // region stage of a level, launched under its (x,theta) stage
    hipStream_t stream4 = nullptr;   // round 6: the drain launch of the queue form of the region stage (beside the early launch on stream3)
    hipEvent_t ev_rjoin2 = nullptr, ev_nth = nullptr;   // its completion; "the theta list's length has been published"
    // Round 6: the streamed one-step dictionaries of a storing level (k_x1) run on stream4 BESIDE the end of their level (children, pruned
    // masks, counters) and the next level's KKT kernel -- nothing of those reads a dictionary; the first kernel of the next level that
    // does waits for ev_x1done (x1_join).  MPC_X1_DEFER=0: in line, as round 5.  Not when a profile asks for per-kernel event times.
    hipEvent_t ev_x1go = nullptr, ev_x1done = nullptr;
    bool x1_pending = false;
    std::function<int()> x1_stash;   // x1_defer >= 2: the deferred k_x1 launch, issued by x1_flush behind the level's last kernels
    int x1_defer = 1;           // 1: k_x1 on its own stream from the plan pass on; 2: issued behind the level's last kernels (measured: the children stage gets its 0.26 ms back, the next level's KKT kernel loses 0.40 beside k_x1 -- its W gathers live in the L2 that k_x1 streams through: config 4 3.89 -> 3.95 ms, config 3 4.17 -> 4.27); 0: in line
    int x1_lds_cap = 0;
    int xq_skip_below = 100000;  // the last level's product-form quick test leaves a list shorter than this to k_x2 when a tableau row is one lane's (MPC_XQ_SKIP_BELOW; 0: never).  Measured: config 3 (62 k left over) 4.26 -> 4.18 ms, config 4 (2.7 k) unchanged -- its last level ends with the region kernel
    int x_second_max = 1024;     // k_x2: a level's budget of doubtful cached runs repeated from D0 in the kernel (MPC_X_SECOND_MAX; 0: all go to the LDS engine)
    double x_fresh_limit = 1e6;   // k_x2: growth up to which a run from D0 decides (DictCache::fresh_limit; MPC_X_FRESH_LIMIT=0: GROWTH_SAFE)
    int kkt_spread_threads = 1 << 19;   // classic path: the spread form while candidates x KKT_SPREAD stays below this (MPC_KKT_SPREAD_THREADS)
    int kkt_spread = 1;         // small levels: k_kkt_thread with KKT_SPREAD lanes per candidate (MPC_KKT_SPREAD=0: one lane)
    int helper_it = 4;          // scan / partition helpers: 4 = four-wavefront workgroups of 4 items per thread, 1 = 1024-thread workgroups (MPC_HELPER_IT)
    // Round 6: the pruned list bucketed by smallest non-equality member (kernels.hpp, k_children_count_b), rebuilt at the start of every level
    // whose children stage is large enough to pay for it (MPC_PRUNED_BUCKET_MIN: parents x pruned sets; 0 = never)
    DevBuf pruned_b, pruned_head;
    double pruned_bucket_min = 2.0e7;
    long long pruned_bucket_np = 1024;   // MPC_PRUNED_BUCKET_NP: ... and at least this many pruned sets (tests: 1)
    int r2_early = 0;                // MPC_R2_EARLY=1: the queue form of a large last level's region stage (measured slower, DESIGN 6h: off; tests switch it on)
    long long r2_early_min = 65536;  // MPC_R2_EARLY_MIN: smallest level that takes the queue form
    int r2_early_wpc = 0, r2_early_spin = 200000, r2_early_thw = 1, r2_early_prio = 3;   // ... MPC_R2_EARLY_THW: theta wavefronts per SIMD beside the early launch; MPC_R2_EARLY_PRIO: their issue priority   // MPC_R2_EARLY_WPC: wavefronts per CU of the early launch; MPC_R2_EARLY_SPIN: looks at an empty queue before a wavefront leaves
    long long n_r2_early = 0;        // regions the early launch built in the last level run (statistics)
    hipEvent_t ev_rfork = nullptr, ev_rjoin = nullptr, ev_rgo = nullptr;
    bool no_roverlap = false;        // MPC_NO_ROVERLAP=1 / mpc_set_region_overlap(h, 0): region stage after the (x,theta) stage (no overlap)
    bool r3_dirty = false;           // a region kernel launched on stream3 has not been joined by a completed level yet
    int test_spare = 0;              // MPC_TEST_SPARE=N (tests): N fewer spare region slots than the overlapped launch would reserve
    int rsplit_max = 4;              // MPC_RSPLIT_MAX: most wavefronts that share one optimal candidate in k_region2 (power of two, <= 16; measured:
                                     // 8 and 16 shorten no level of config 4 or 2 -- every wavefront repeats the row build and the Chebyshev LP, 40 % of
                                     // a region at 4 -- and move the facet list of one sliver region)
    bool no_lean = false;            // MPC_NO_LEAN=1: large levels read every list length back (round-2 behaviour); default: only the lengths the
                                     // host needs to size the region stage are read back, the other stages take theirs from device memory
    bool theta_open = false;         // the parameter set is open in some direction (or the program has equality rows only): the reference's
                                     // optimality LP can be unbounded -> k_recession behind every verdict stage, no overlapped region launch,
                                     // no level without host round trips, no shared launches (MPC_NO_RECESSION=1: round-3 behaviour, A/B)
    int xqg_per_cu = XQG_WAVES;      // MPC_XQG_PER_CU: workgroups (of four wavefronts) per CU of the persistent k_xq_grouped launch
    int x2_wpc = 12, x2_div = 16, xq_wpc = 20;    // MPC_X2_WPC (most) / MPC_X2_DIV (items per wavefront) / MPC_XQ_WPC: wavefronts per CU of the persistent
                                     // k_x2 / k_xq launches (round 3: 16 and 32 whatever the size of the level)
    std::vector<std::pair<void *, size_t>> parked_blocks;   // device blocks the level buffers have outgrown (DevBuf::parked), given back by graveyard_flush
    bool region_side_stream = false; // this level's k_region2 launch ran on the side stream, under the (x,theta) stage (mpc_level_stats)
    bool r3_fork_event = false;      // MPC_R3_FORK=1: the region stream starts behind an event of the main stream (round-3 form; A/B)
    int r2_cap_pct = 100;            // MPC_R2_CAP: share (per cent) of k_region2's wave slots an overlapped one-wave-per-candidate launch may take
    bool no_fetch_kernel = false;    // MPC_NO_FETCH_KERNEL=1: the solve loop fetches the records of a level that did not stream with copy commands and waits (A/B)
    bool no_spec_tail = false;       // MPC_NO_SPEC_TAIL=1: a large level waits for the second partition and for the region kernel's give-up count before
                                     // it queues its end (round-3 behaviour); default: the end is queued behind the partition, one synchronisation
    bool test_small_fallback = false; // MPC_TEST_SMALL_FALLBACK=1 (tests): every level run without host round trips reports "repeat on the classic path"
    bool timing = true;              // mpc_set_timing: HIP-event records around the stages and kernels of a level (off: their times in the stats are 0)
    bool no_kev = false;             // MPC_NO_KEV=1: never
    bool no_smallpath = false;       // MPC_NO_SMALLPATH=1: levels of any size take the classic path with its host round trips (A/B)
    long long smallpath_max = 4096;  // MPC_SMALLPATH_MAX: largest level (candidates) that runs without host round trips (measured: config 4 is
                                     // fastest with 1,024-4,096 -- a level of 15,691 candidates prefers the classic path, which streams its records)
    bool x_first = true;             // MPC_NO_X_FIRST=1: the (x,theta) stage of a storing level waits for the region stage's launch as in round 4 (A/B)
    long long x_first_min = 4096, x_first_max = 65536;   // MPC_X_FIRST_MIN / _MAX: levels (candidates) whose (x,theta) stage is queued before the read-back.  Measured, config 4: level 3
                                     // (9,880 candidates) 0.72 -> 0.67 ms; level 4 (181 k) 1.51 -> 1.58: there the stage fills the GPU before the region kernel's long wavefronts are placed
                                     // (region kernel 0.48 -> 0.60 ms, the stage itself 0.93 -> 1.03) -- large levels keep the region launch first
    hipEvent_t ev_part = nullptr;
    bool no_batch_plans = true;      // MPC_BATCH_PLANS=1 switches the one-step plans on in the shared launches (tests).  Measured on the bench enumeration (64 sub-programs, 16-column
                                     // records of 3.5 KB) and on 128 small programs: SLOWER with them (121 against 108 ms; 20.9 against 19.6 ms, device 10.1 against 8.6) -- the register
                                     // simplex on so small a record costs less than the plan pass's batch of dependent look-ups; off by default
    long long batch_plan_min = 64;   // MPC_BATCH_PLAN_MIN: smallest member level (candidates) that plans
    bool no_kkt_lists = false;       // MPC_NO_KKT_LISTS=1: the work lists behind k_kkt_thread by compaction of the status array (round 4; A/B, tests)
    bool no_small_rx = false;        // MPC_NO_SMALL_RX=1: region kernel and (x,theta) kernel of a small level as two launches (A/B, tests)
    bool no_small_fuse = false;      // MPC_NO_SMALL_FUSE=1: the small path with its round-4 launches (doubtful candidates re-solved in place; A/B, tests)
    long long n_smallpath_doubtful = 0;   // small levels repeated because the fused form met a doubtful candidate
    long long n_smallpath = 0, n_smallpath_fallback = 0;   // levels run that way / of which repeated on the classic path
    DevBuf dcnt;                     // device-resident list lengths of such a level
    long long roverlap_min = 2048, roverlap_long = 50000;   // MPC_ROVERLAP_MIN / MPC_ROVERLAP_LONG (items of the (x,theta) stage)
    int wall_khz = 100000;           // rate of wall_clock64() on the device
    int test_late = 0;               // MPC_TEST_LATE=N (tests): the overlapped launch leaves N optimal candidates to the late path
    bool own_stream = false;
    int n_cu = 256;
    std::string error;
    std::mutex em;                   // fail() is called by the worker thread and by the caller's thread (mpc_level_chunk_wait)
    // problem
    int n_x = 0, n_t = 0, n_c = 0, n_eq = 0, n_tc = 0, is_qp = 0, kkt_mode = 0;
    int mw = MPC_MASK_WORDS;   // 64-bit words of an active-set mask: 2 (n_c <= 128) or 4 (n_c <= 256)
    DevBuf blocks;            // all read-only problem blocks, one allocation
    DevBuf iblocks;           // integer blocks (row / column maps of the pre-crashed dictionary)
    DevProblem Pv{}, Pr{};    // verdict / region kernel views (same blocks, different LDS layouts)
    int lds_v = 0, lds_r = 0; // dynamic LDS bytes per wavefront
    int debug_cycles = 0;     // MPC_DEBUG_CYCLES=1: per-level cycle breakdown on stderr
    int force_xqgroup = 0;    // MPC_FORCE_XQGROUP=1: k_xq_grouped whatever the number of siblings (tests)
    long long last_level_n = 0;   // candidates of the previous level (= the parents of this one)
    int no_xqgroup = 0;       // MPC_NO_XQGROUP=1: the last level's quick test reads the parent records from HBM per candidate (A/B)
    int no_rbox = 0;          // MPC_NO_RBOX=1: no bounding-box screen of the region rows in k_region2 (A/B)
    int no_rsplit = 0;        // MPC_NO_RSPLIT=1: one wavefront per candidate in k_region2 whatever the load (A/B)
    int no_xquick = 0;        // MPC_NO_XQUICK=1: no quick (x,theta) test on the last level (A/B)
    int x1 = 1;               // MPC_X1=0: every dictionary of a storing level by the register simplex k_x2 (rounds 1-4); 1: one-step plans (k_xq_thread, plan mode) streamed by k_x1, from the generating parent only; 2 (default): ... and from the candidate's other parents
    long long xqt_min = 4096, x1_min = 2048;   // MPC_XQT_MIN / MPC_X1_MIN: smallest list the one-thread pass / the one-step plans take (tests set 1: every level of a small program goes through them)
    int x1_wpc = 16;          // MPC_X1_WPC: wavefronts per CU of k_x1 (config 4, level 4, beside the region kernel: x stage 1.13 / 0.95 / 1.02 ms with 8 / 16 / 32)
    long long n_x1 = 0;       // dictionaries of the last level run that k_x1 wrote
    float ms_x1 = 0;
    int xq_retry = 0;         // MPC_XQ_RETRY=1: a doubtful pivot met by the quick test is flagged by the quick test itself and re-solved at once on the second stream (round 5; off: on config 3 half of the doubtful candidates only show in k_x2, beyond the quick test's sixteen iterations, so the level pays the LDS engine twice -- 3.55 ms against 3.28)
    int no_xq_early = 0;      // MPC_NO_XQ_EARLY=1: the thread pass of the quick test always behind the theta stage, -1: always beside it (A/B)
    long long prev_regions = 0, xq_early_regions = 0;   // MPC_XQ_EARLY_REGIONS = r > 0: the pass runs beside the theta stage only when the level before found fewer than r regions (0: always)
    int xqt_wpc = 16;         // MPC_XQT_WPC: wavefronts per CU of k_xq_thread (it is bound by the cache's request rate: config 4's level 0.45 ms alone with 8 per CU, 0.49 with 24; beside the region kernel 0.92 / 0.70 / 0.78 / 0.75 with 4 / 8 / 12 / 16)
    int xqg_overlap = 0;      // MPC_XQG_OVERLAP=1: the region stage runs under the (x,theta) stage also when the quick test is the grouped one (experiment)
    int xq_thread = -1;       // MPC_XQ_THREAD: 0 = the quick test without its one-thread-per-candidate first pass k_xq_thread (round 5); 1 = the pass against the generating parent only; n >= 2 = ... and up to n - 1 other parents; default: every other parent
    long long n_xq_thread = 0; float ms_xq_thread = 0;   // the last level run: candidates that pass decided, its time
    bool fetch_nowait = false; // mpc_level_regions_slots_nowait: even the integer heads are only queued
    bool skip_small = false;  // mpc_level_run_batch: this level already went through the no-round-trip launches and has to be repeated classically
    size_t o_elim[6] = {0, 0, 0, 0, 0, 0};   // offsets of Wr, UVr, AATr, Me, Ne, gE in `blocks` (mpc_program_block)
    bool elim_ok = false;     // the blocks with the equality rows eliminated are valid
    int no_kkt_thread = 0;    // MPC_NO_KKT_THREAD=1: KKT solves stay inside the wave kernels (A/B)
    int force_v1 = 0;         // MPC_FORCE_V1=1 in the environment: never use k_verdict2 (A/B comparisons, tests)
    int fast = 0;             // 1: register-engine kernels (k_theta2 / k_x2) with k_verdict as the retry path
    int fast_t = 0, fast_x = 0; // instantiation selectors
    long long n_needx = 0;
    DevProblem Pf{};          // view for k_verdict2 (small LDS layout: no tableau)
    int lds_f = 0, grid_f = 0;
    DevBuf retry_list, theta_list, vretry_list, status_tmp, part_counts, part_lists, kept_g, done_g, pf_dev, pr2_dev, headd, headi, epool, facet_flags, kkt_code, kkt_L, theta_blocks, xq_groups, xq_list, x1_buf, xretry_list;
    ThetaArgs targs{};
    // (x,theta) dictionary cache: [0]/[1] ping-pong between the level being read (parents) and the level being written
    DevBuf dict_d[2], dict_i[2], dict_stored[2], parent_slot, parent_slot_next;
    int dict_cur = 0;
    bool have_prev_dict = false, have_parent_slot = false, storing = false;
    long long dict_stride_d = 0, dict_stride_i = 0;
    double dict_budget_gb = 48.0;

    DevProblem Pr2{};         // view for k_region2
    int lds_r2 = 0, grid_r2 = 0, fast_r = -1;   // fast_r: k_region2 instantiation, -1 = none (n_t == 1 or too many rows)
    bool used_region2 = false;
    long long rretry_rows = -1;      // >= 0: the re-solved candidates' records have been written into their slots on the device (k_rretry_merge); rows of epool in use then (bound)
    long long n_rretry = 0, n_erows = 0;
    int fd = 0, fi = 0;
    HostBuf st_list, st_status, st_hd, st_hi, st_pool, st_fxd, st_fxi, st_rlist;   // pinned staging for region fetches
    int grid_v = 0, grid_r = 0;
    long long rec_d = 0, rec_i = 0;
    // frontier / pruned
    int32_t *tot_host = nullptr, *tot_dev = nullptr;   // 16 counters in pinned host memory that the kernels write directly: list lengths need no copy
    const int32_t *opt_ptr = nullptr;                  // list of the optimal candidates of the level (a view, not a copy)
    DevBuf frontier, children, status, pruned, flag, pos, opt_list, childmask, count, offset, recd, reci, ctr, scratch, sums;
    long long n = 0;
    long long n_prev = 0;     // candidates of the level before (0: unknown -- frontier set by the caller)
    int k = 0;
    long long n_pruned = 0;
    long long n_pruned_extra = 0;   // masks added from outside (other ranks') behind this level's own new ones, before mpc_frontier_advance
    // level results
    bool level_done = false;
    long long n_opt = 0, n_children = 0, n_pruned_new = 0, n_regions = 0;
    // ---- region records streamed to the host while the region kernel runs (mpc_level_start with MPC_LEVEL_STREAM) ----
    struct StreamOut {
        void *hd = nullptr, *hi = nullptr, *er = nullptr;   // page-locked blocks the region kernel writes (zero-copy)
        long long n_slots = 0, cap_rows = 0;
        int shift = 8, n_chunks = 0;
        bool active = false;    // this level's records are in these blocks, not in headd / headi / epool
        bool taken = false;     // the caller owns the blocks (mpc_level_stream_info handed them over)
    } so;
    HostBuf st_flags;           // chunk flags (host memory the kernel writes)
    int cw_chunks = 0;          // chunks of the most recent streamed level (mpc_level_chunk_wait keeps working while the worker
                                // already runs the base-set check behind that level, which resets `so`)
    DevBuf chunk_count;
    // ---- worker thread behind mpc_level_start / mpc_level_wait ---------------------------------------------------------
    std::thread worker;
    std::mutex wm;
    std::condition_variable wcv;
    int w_req = 0;              // 0 idle, 1 run a level, 2 exit
    bool w_busy = false, w_stream_ready = false;
    // mirrors of the three flags for a short spin before the condition-variable wait: a hand-over through the condition variable
    // alone costs 20-50 us of GPU idle time per level (tools/gpu_idle.sh), a spinning waiter sees the flag within a microsecond
    std::atomic<int> a_req{0}, a_busy{0}, a_ready{0};
    int w_gen = 0, w_flags = 0, w_rc = 0;
    // result of the base-set check the worker ran behind the last level (MPC_LEVEL_THEN_BASE)
    bool base_valid = false;
    uint8_t base_status = 0;
    long long base_regions = 0;
    std::vector<double> base_rec_d;
    std::vector<int32_t> base_rec_i;
    mpc_level_stats w_stats{};
    // ---- the whole level loop on the worker thread (mpc_solve_start / _level / _chunk_wait / _level_wait / _wait) -----------------
    struct SolveLevel {
        int32_t k = 0, mode = 0, chunk = 0, n_chunks = 0;   // mode: 0 nothing to read, 1 streamed in chunks, 2 complete arrays
        int64_t n = 0, n_slots = 0, n_rows = 0;
        void *hd = nullptr, *hi = nullptr, *er = nullptr;   // page-locked blocks; the caller's once mpc_solve_level has handed them over
        bool handed = false;
        const int32_t *flags = nullptr;                     // chunk flags (valid while the level runs)
        const int32_t *ready_flag = nullptr;                // mode 2, copied by k_fetch_slots: raised (behind head_i, in its block) when the arrays are complete
        mpc_level_stats stats{};
        double ms_wall = 0.0;
        std::atomic<int> ready{0}, done{0};
    };
    std::unique_ptr<SolveLevel[]> sv_levels;                // MPC_MAX_NC + 2 entries, allocated by the first mpc_solve_start
    int sv_max_levels = 0, sv_flags = 0, sv_rc = 0;
    int sv_cur = -1;                                        // level the worker is running (-1: not inside a solve loop)
    std::atomic<int> sv_n{0}, sv_finished{1};               // levels begun so far; the loop has ended
    // ---- connected-graph traversal: wave, visited set and pending neighbours resident on the device (graph.hpp) ----------
    struct GraphState {
        bool active = false;
        int variant = 0;                 // 0 combinatorial_graph, 1 graph
        DevBuf wave, visited, pending, tmp_a, tmp_b, sort_tmp, card, idx, card2, idx2, facet, cnt, off, hist;
        long long n_wave = 0, n_visited = 0, n_pending = 0;
        std::vector<int> gk;             // groups of the current wave: cardinality,
        std::vector<long long> goff, gcnt;   // first mask and number of masks
    } g;
    hipEvent_t ev_hi = nullptr;   // completion of the head_i copy of an asynchronous slot fetch
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t kev[14] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // around k_theta2 / the main k_x2 launch / k_region2 / k_kkt_thread / k_xq / k_xq_thread
};

namespace {

int fail(mpc_handle *h, int code, const std::string &msg) {
    if (h) { std::lock_guard<std::mutex> lk(h->em); h->error = msg; } else g_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            return fail(h, MPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));               \
    } while (0)

// dense host helpers (one-off program setup; fp64)
bool cholesky(std::vector<double> &M, int n) {
    double dmax = 0;
    for (int i = 0; i < n; ++i) dmax = std::max(dmax, std::fabs(M[(size_t)i * n + i]));
    for (int j = 0; j < n; ++j) {
        double d = M[(size_t)j * n + j];
        for (int l = 0; l < j; ++l) d -= M[(size_t)j * n + l] * M[(size_t)j * n + l];
        if (!(d > 1e-10 * dmax)) return false;
        const double s = std::sqrt(d);
        M[(size_t)j * n + j] = s;
        for (int i = j + 1; i < n; ++i) {
            double v = M[(size_t)i * n + j];
            for (int l = 0; l < j; ++l) v -= M[(size_t)i * n + l] * M[(size_t)j * n + l];
            M[(size_t)i * n + j] = v / s;
        }
    }
    return true;
}
// x := Q^-1 x given the lower Cholesky factor
void chol_apply(const std::vector<double> &L, int n, double *x) {
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        for (int l = 0; l < i; ++l) v -= L[(size_t)i * n + l] * x[l];
        x[i] = v / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double v = x[i];
        for (int l = i + 1; l < n; ++l) v -= L[(size_t)l * n + i] * x[l];
        x[i] = v / L[(size_t)i * n + i];
    }
}

struct Layout { int off_T, off_K, off_L, off_E, off_X, n_doubles, off_as, off_inact, off_colvar, off_rowvar, off_rowkind, off_kept, off_pri, off_stored, n_ints, bytes; };

Layout make_layout(int size_T, int size_K, int size_L, int size_E, int size_X, int kmax, int n_c, int ld_max, int rows_max, int rows_t) {
    Layout l{};
    int o = 0;
    auto take = [&](int sz) { int r = o; o += (sz + 1) & ~1; return r; };
    l.off_T = take(size_T); l.off_K = take(size_K); l.off_L = take(size_L); l.off_E = take(size_E); l.off_X = take(size_X);
    l.n_doubles = o;
    int q = 0;
    auto itake = [&](int sz) { int r = q; q += sz; return r; };
    l.off_as = itake(kmax + 1); l.off_inact = itake(n_c + 1); l.off_colvar = itake(ld_max + 2);
    l.off_rowvar = itake(rows_max + 3); l.off_rowkind = itake(rows_max + 3); l.off_kept = itake(rows_t + 1); l.off_pri = itake(rows_max + 3); l.off_stored = itake(rows_t + 1);
    l.n_ints = q;
    l.bytes = (l.n_doubles * 8 + l.n_ints * 4 + 15) & ~15;
    return l;
}

void apply_layout(DevProblem &P, const Layout &l) {
    P.off_T = l.off_T; P.off_K = l.off_K; P.off_L = l.off_L; P.off_E = l.off_E; P.off_X = l.off_X; P.n_doubles = l.n_doubles;
    P.off_as = l.off_as; P.off_inact = l.off_inact; P.off_colvar = l.off_colvar; P.off_rowvar = l.off_rowvar;
    P.off_rowkind = l.off_rowkind; P.off_kept = l.off_kept; P.off_pri = l.off_pri; P.off_stored = l.off_stored; P.n_ints = l.n_ints;
}

int waves_per_cu(int lds_bytes) { return std::max(1, std::min(16, (160 * 1024) / std::max(lds_bytes, 1))); }

// LU with partial pivoting of the n x n matrix M (row major, overwritten); perm receives the row order.
bool lu_factor(std::vector<double> &M, int n, std::vector<int> &perm) {
    perm.resize(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    double scale = 0;
    for (double v : M) scale = std::max(scale, std::fabs(v));
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int i = c + 1; i < n; ++i) if (std::fabs(M[(size_t)i * n + c]) > std::fabs(M[(size_t)p * n + c])) p = i;
        if (!(std::fabs(M[(size_t)p * n + c]) > 1e-12 * scale)) return false;
        if (p != c) { for (int j = 0; j < n; ++j) std::swap(M[(size_t)p * n + j], M[(size_t)c * n + j]); std::swap(perm[p], perm[c]); }
        for (int i = c + 1; i < n; ++i) {
            const double f = M[(size_t)i * n + c] / M[(size_t)c * n + c];
            M[(size_t)i * n + c] = f;
            if (f != 0.0) for (int j = c + 1; j < n; ++j) M[(size_t)i * n + j] -= f * M[(size_t)c * n + j];
        }
    }
    return true;
}
void lu_solve_host(const std::vector<double> &LU, const std::vector<int> &perm, int n, const double *rhs, double *x) {
    for (int i = 0; i < n; ++i) { double v = rhs[perm[i]]; for (int l = 0; l < i; ++l) v -= LU[(size_t)i * n + l] * x[l]; x[i] = v; }
    for (int i = n - 1; i >= 0; --i) { double v = x[i]; for (int l = i + 1; l < n; ++l) v -= LU[(size_t)i * n + l] * x[l]; x[i] = v / LU[(size_t)i * n + i]; }
}

}  // namespace

// the buffers a level (re)sizes by its number of candidates: what they outgrow is parked (DevBuf::parked) and given back to the pool by
// graveyard_flush, which the level paths call behind their closing synchronisation (all streams of the handle have been joined by then)
static std::vector<DevBuf *> level_buffers(mpc_handle *h) {
    return {&h->frontier, &h->children, &h->status, &h->pruned, &h->flag, &h->pos, &h->opt_list, &h->childmask, &h->count, &h->offset, &h->recd, &h->reci,
            &h->sums, &h->retry_list, &h->headd, &h->headi, &h->epool, &h->facet_flags, &h->kkt_code, &h->kkt_L, &h->xq_groups, &h->xq_list, &h->x1_buf, &h->xretry_list, &h->theta_list, &h->vretry_list,
            &h->status_tmp, &h->part_counts, &h->part_lists, &h->kept_g, &h->done_g, &h->dict_d[0], &h->dict_d[1], &h->dict_i[0], &h->dict_i[1],
            &h->dict_stored[0], &h->dict_stored[1], &h->parent_slot, &h->parent_slot_next};
}
static void graveyard_flush(mpc_handle *h) {
    for (auto &b : h->parked_blocks) dev_pool_give(b.first, b.second);
    h->parked_blocks.clear();
}
static int create_fill(const mpc_problem *p, int32_t device, void *stream, mpc_handle *h);
static void stream_release(mpc_handle *h);
static void solve_release(mpc_handle *h);
static double batch_level_gb(const mpc_handle *h, int32_t gen_children);
static int fetch_many_wait(int device);
static int lp_batch_impl(int32_t device, int64_t n_lp, int32_t m, int32_t n, const double *A, int32_t shared_A, const double *b,
                         int32_t shared_b, const double *c, int32_t shared_c, const uint8_t *eq, int32_t *status, double *x,
                         double *obj, int32_t *iters, int32_t *tight);

extern "C" {

int mpc_device_count(void) { return device_count_cached(); }

// the build stamp says which binary a record was made with: the compiler's date / time of THIS translation unit and the hipcc version
#define MPC_STR2(x) #x
#define MPC_STR(x) MPC_STR2(x)
const char *mpc_version(void) {
    return "mpcombi-hip 0.4 (gfx950; built " __DATE__ " " __TIME__ ", hip " MPC_STR(HIP_VERSION_MAJOR) "." MPC_STR(HIP_VERSION_MINOR) "." MPC_STR(HIP_VERSION_PATCH) ")";
}
const char *mpc_last_global_error(void) { return g_error.c_str(); }
const char *mpc_last_error(const mpc_handle *h) { return h ? h->error.c_str() : g_error.c_str(); }

int mpc_create(const mpc_problem *p, int32_t device, void *stream, mpc_handle **out) {
    if (!p || !out) return fail(nullptr, MPC_ERR_INVALID, "null argument");
    *out = nullptr;
    const int nx = p->n_x, nt = p->n_t, nc = p->n_c, ne = p->n_eq, ntc = p->n_tc;
    if (nx < 1 || nt < 1 || nc < 1 || ne < 0 || ne > nc || ntc < 0) return fail(nullptr, MPC_ERR_INVALID, "bad dimensions");
    if (nc > MPC_MAX_NC) return fail(nullptr, MPC_ERR_INVALID, "n_c > 256 is not supported (active sets are bit masks of at most four words)");
    if (nc + ntc + 2 > MPC_MAX_ROWS * 4) return fail(nullptr, MPC_ERR_INVALID, "too many rows");
    if (nt > 64 || nx > 256) return fail(nullptr, MPC_ERR_INVALID, "n_t > 64 or n_x > 256 is not supported");
    if (!p->A || !p->b || !p->F || !p->c || !p->H || (ntc > 0 && (!p->A_t || !p->b_t))) return fail(nullptr, MPC_ERR_INVALID, "null matrix");
    int ndev = 0;
    if ((ndev = device_count_cached()) < 1) return fail(nullptr, MPC_ERR_HIP, "no HIP device available (libmpcombi_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, MPC_ERR_INVALID, "device index out of range");
    // every failure from here on goes through one cleanup path: mpc_destroy returns the handle's streams, events, device
    // blocks and pinned block to the pools (a caller that creates one handle per binary fixation must not leak on OOM)
    mpc_handle *h = new mpc_handle();
    for (DevBuf *b : level_buffers(h)) b->parked = &h->parked_blocks;
    const int rc_fill = create_fill(p, device, stream, h);
    if (rc_fill != MPC_OK) { (void)mpc_destroy(h); return rc_fill; }
    *out = h;
    return MPC_OK;
}

}  // extern "C"

// fills a fresh handle; on failure the caller destroys it (error text in the thread-local g_error, see mpc_last_global_error)
static int create_fill(const mpc_problem *p, int32_t device, void *stream, mpc_handle *h) {
    const int nx = p->n_x, nt = p->n_t, nc = p->n_c, ne = p->n_eq, ntc = p->n_tc;
    h->device = device;
    HIP_TRY(nullptr, hipSetDevice(device));
    h->n_cu = cu_count(device);
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) h->wall_khz = khz; }
    if (stream) { h->stream = reinterpret_cast<hipStream_t>(stream); h->own_stream = false; }
    else { HIP_TRY(nullptr, pooled_stream(&h->stream)); h->own_stream = true; }
    for (auto &e : h->ev) HIP_TRY(nullptr, pooled_event(&e, true));
    for (auto &e : h->kev) HIP_TRY(nullptr, pooled_event(&e, true));
    HIP_TRY(nullptr, pooled_stream(&h->stream2));
    HIP_TRY(nullptr, pooled_event(&h->ev_hi, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_fork, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_join, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_xfork, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_part, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_xjoin, false));
    HIP_TRY(nullptr, pooled_stream(&h->stream3));
    HIP_TRY(nullptr, pooled_stream(&h->stream4));
    HIP_TRY(nullptr, pooled_event(&h->ev_rjoin2, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_nth, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_x1go, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_x1done, false));
    { const char *ev = std::getenv("MPC_X1_DEFER"); if (ev) h->x1_defer = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_X1_LDS_CAP"); if (ev) h->x1_lds_cap = std::max(0, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_KKT_SPREAD"); if (ev) h->kkt_spread = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_KKT_SPREAD_THREADS"); if (ev) h->kkt_spread_threads = std::max(0, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_X_FRESH_LIMIT"); if (ev) h->x_fresh_limit = std::atof(ev); }
    { const char *ev = std::getenv("MPC_X_SECOND_MAX"); if (ev) h->x_second_max = std::max(0, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_XQ_SKIP_BELOW"); if (ev) h->xq_skip_below = std::max(0, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_HELPER_IT"); if (ev) h->helper_it = std::atoi(ev) == 1 ? 1 : 4; }
    { const char *ev = std::getenv("MPC_PRUNED_BUCKET_MIN"); if (ev) h->pruned_bucket_min = std::atof(ev); }
    { const char *ev = std::getenv("MPC_PRUNED_BUCKET_NP"); if (ev) h->pruned_bucket_np = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_R2_EARLY"); if (ev) h->r2_early = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_R2_EARLY_MIN"); if (ev) h->r2_early_min = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_R2_EARLY_WPC"); if (ev) h->r2_early_wpc = std::max(0, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_R2_EARLY_SPIN"); if (ev) h->r2_early_spin = std::max(0, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_R2_EARLY_THW"); if (ev) h->r2_early_thw = std::max(1, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_R2_EARLY_PRIO"); if (ev) h->r2_early_prio = std::max(0, std::min(3, std::atoi(ev))); }
    HIP_TRY(nullptr, pooled_event(&h->ev_rfork, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_rjoin, false));
    HIP_TRY(nullptr, pooled_event(&h->ev_rgo, false));
    h->n_x = nx; h->n_t = nt; h->n_c = nc; h->n_eq = ne; h->n_tc = ntc; h->is_qp = p->Q != nullptr;
    h->mw = nc <= 64 * MPC_MASK_WORDS ? MPC_MASK_WORDS : 4;
    { const char *ev = std::getenv("MPC_FORCE_V1"); h->force_v1 = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_DEBUG_CYCLES"); h->debug_cycles = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_RSPLIT"); h->no_rsplit = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_RBOX"); h->no_rbox = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_XQGROUP"); h->no_xqgroup = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_FORCE_XQGROUP"); h->force_xqgroup = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_XQUICK"); h->no_xquick = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_XQ_THREAD"); h->xq_thread = ev ? std::atoi(ev) : -1; }
    { const char *ev = std::getenv("MPC_XQG_OVERLAP"); h->xqg_overlap = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_XQT_WPC"); if (ev && std::atoi(ev) > 0) h->xqt_wpc = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_X1"); h->x1 = ev ? std::atoi(ev) : 2; }
    { const char *ev = std::getenv("MPC_XQ_RETRY"); h->xq_retry = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_X1_WPC"); if (ev && std::atoi(ev) > 0) h->x1_wpc = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_XQT_MIN"); if (ev) h->xqt_min = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_X1_MIN"); if (ev) h->x1_min = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_NO_XQ_EARLY"); h->no_xq_early = ev ? std::atoi(ev) : 0; }
    { const char *ev = std::getenv("MPC_XQ_EARLY_REGIONS"); if (ev) h->xq_early_regions = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_NO_KKT_THREAD"); h->no_kkt_thread = ev ? std::atoi(ev) : 0; }   // 2: only the small-level path
    { const char *ev = std::getenv("MPC_NO_ROVERLAP"); h->no_roverlap = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_TEST_LATE"); h->test_late = ev ? std::atoi(ev) : 0; }
    { const char *ev = std::getenv("MPC_TEST_SPARE"); h->test_spare = ev ? std::atoi(ev) : 0; }
    { const char *ev = std::getenv("MPC_NO_KEV"); h->no_kev = ev && ev[0] == '1'; h->timing = !h->no_kev; }
    { const char *ev = std::getenv("MPC_NO_X_FIRST"); h->x_first = !(ev && ev[0] == '1'); }
    { const char *ev = std::getenv("MPC_X_FIRST_MIN"); if (ev) h->x_first_min = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_X_FIRST_MAX"); if (ev) h->x_first_max = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_BATCH_PLANS"); h->no_batch_plans = !(ev && ev[0] == '1'); }
    { const char *ev = std::getenv("MPC_BATCH_PLAN_MIN"); if (ev) h->batch_plan_min = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_NO_KKT_LISTS"); h->no_kkt_lists = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_SMALL_RX"); h->no_small_rx = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_SMALL_FUSE"); h->no_small_fuse = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_SMALLPATH"); h->no_smallpath = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_LEAN"); h->no_lean = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_SPEC_TAIL"); h->no_spec_tail = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_NO_FETCH_KERNEL"); h->no_fetch_kernel = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_X2_WPC"); if (ev && std::atoi(ev) > 0) h->x2_wpc = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_XQ_WPC"); if (ev && std::atoi(ev) > 0) h->xq_wpc = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_XQG_PER_CU"); if (ev && std::atoi(ev) > 0) h->xqg_per_cu = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_X2_DIV"); if (ev && std::atoi(ev) > 0) h->x2_div = std::atoi(ev); }
    { const char *ev = std::getenv("MPC_R3_FORK"); h->r3_fork_event = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_R2_CAP"); if (ev && std::atoi(ev) > 0) h->r2_cap_pct = std::min(100, std::atoi(ev)); }
    { const char *ev = std::getenv("MPC_TEST_SMALL_FALLBACK"); h->test_small_fallback = ev && ev[0] == '1'; }
    { const char *ev = std::getenv("MPC_RSPLIT_MAX"); if (ev) { int v = std::atoi(ev); h->rsplit_max = v >= 16 ? 16 : (v >= 8 ? 8 : (v >= 4 ? 4 : (v >= 2 ? 2 : 1))); } }
    { const char *ev = std::getenv("MPC_SMALLPATH_MAX"); if (ev) h->smallpath_max = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_ROVERLAP_MIN"); if (ev) h->roverlap_min = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_ROVERLAP_LONG"); if (ev) h->roverlap_long = std::atoll(ev); }
    { const char *ev = std::getenv("MPC_DICT_BUDGET_GB"); if (ev) h->dict_budget_gb = std::atof(ev); }

    const int nr = nt + 1;
    // ---- host-side one-off blocks ----------------------------------------------------------------------
    // W = A Q^-1 A', UV, G', X0H and A A' are formed on the DEVICE by the MFMA set-up kernel (setup_mfma.hip) once the raw
    // matrices are in HBM (below).  MPC_HOST_SETUP=1 keeps the scalar host computation instead (A/B comparisons, tests).
    const bool host_setup = [] { const char *ev = std::getenv("MPC_HOST_SETUP"); return ev && ev[0] == '1'; }();
    std::vector<double> W, UV, Gt, X0H;
    int mode = 1;
    if (p->Q && host_setup) {
        std::vector<double> L(p->Q, p->Q + (size_t)nx * nx);
        for (int i = 0; i < nx; ++i) for (int j = 0; j < i; ++j) { const double s = 0.5 * (L[(size_t)i * nx + j] + L[(size_t)j * nx + i]); L[(size_t)i * nx + j] = s; L[(size_t)j * nx + i] = s; }
        if (cholesky(L, nx)) {
            mode = 0;
            Gt.assign((size_t)nc * nx, 0.0);
            for (int i = 0; i < nc; ++i) { std::copy(p->A + (size_t)i * nx, p->A + (size_t)(i + 1) * nx, Gt.begin() + (size_t)i * nx); chol_apply(L, nx, &Gt[(size_t)i * nx]); }
            W.assign((size_t)nc * nc, 0.0);
            for (int i = 0; i < nc; ++i) for (int j = 0; j <= i; ++j) {
                double s = 0; for (int l = 0; l < nx; ++l) s += p->A[(size_t)i * nx + l] * Gt[(size_t)j * nx + l];
                W[(size_t)i * nc + j] = s; W[(size_t)j * nc + i] = s;
            }
            // columns of [c | H] through Q^-1
            std::vector<double> QiCH((size_t)nx * nr), col(nx);
            for (int t = 0; t < nr; ++t) {
                for (int i = 0; i < nx; ++i) col[i] = t == 0 ? p->c[i] : p->H[(size_t)i * nt + t - 1];
                chol_apply(L, nx, col.data());
                for (int i = 0; i < nx; ++i) QiCH[(size_t)i * nr + t] = col[i];
            }
            X0H.assign((size_t)nx * nr, 0.0);
            for (size_t i = 0; i < X0H.size(); ++i) X0H[i] = -QiCH[i];
            UV.assign((size_t)nc * nr, 0.0);
            for (int i = 0; i < nc; ++i) for (int t = 0; t < nr; ++t) {
                double s = 0; for (int l = 0; l < nx; ++l) s += p->A[(size_t)i * nx + l] * QiCH[(size_t)l * nr + t];
                UV[(size_t)i * nr + t] = s + (t == 0 ? p->b[i] : p->F[(size_t)i * nt + t - 1]);
            }
        }
    }
    const int cols = 1 + nx + nt, rows_x = nc + ntc;
    std::vector<double> base((size_t)rows_x * cols, 0.0);
    for (int i = 0; i < nc; ++i) {
        base[(size_t)i * cols] = p->b[i];
        for (int j = 0; j < nx; ++j) base[(size_t)i * cols + 1 + j] = p->A[(size_t)i * nx + j];
        for (int j = 0; j < nt; ++j) base[(size_t)i * cols + 1 + nx + j] = -p->F[(size_t)i * nt + j];
    }
    for (int i = 0; i < ntc; ++i) {
        base[(size_t)(nc + i) * cols] = p->b_t[i];
        for (int j = 0; j < nt; ++j) base[(size_t)(nc + i) * cols + 1 + nx + j] = p->A_t[(size_t)i * nt + j];
    }
    // ---- pre-crashed dictionary D0 of the (x,theta) LP (see xtheta_from_vertex in kernels.hpp) --------------------
    // A feasible vertex of the base polytope {A x - F theta <= b (first n_eq rows =), A_t theta <= b_t} is found once by
    // the device simplex; the dictionary at that vertex is then rebuilt from the original rows by a fresh LU, so its
    // entries carry factorisation-level accuracy, not the accumulated error of the pivot sequence.
    std::vector<double> d0;
    std::vector<int> d0_rows, d0_cols;
    {
        const int nv = nx + nt;
        std::vector<double> Alp((size_t)rows_x * nv), blp(rows_x);
        for (int i = 0; i < rows_x; ++i) { blp[i] = base[(size_t)i * cols]; for (int j = 0; j < nv; ++j) Alp[(size_t)i * nv + j] = base[(size_t)i * cols + 1 + j]; }
        std::vector<uint8_t> eqf(rows_x, 0);
        for (int i = 0; i < ne; ++i) eqf[i] = 1;
        std::vector<int32_t> tight(rows_x, 0);
        int32_t lp_status = -1;
        int rc0 = rows_x > nv ? lp_batch_impl(device, 1, rows_x, nv, Alp.data(), 1, blp.data(), 1, nullptr, 1, eqf.data(), &lp_status, nullptr, nullptr, nullptr, tight.data()) : MPC_ERR_INVALID;
        HIP_TRY(nullptr, hipSetDevice(device));
        std::vector<int> B;
        for (int i = 0; i < rows_x; ++i) if (tight[i]) B.push_back(i);
        bool ok = rc0 == MPC_OK && lp_status == LP_OPTIMAL && (int)B.size() == nv;
        std::vector<double> MT;
        std::vector<int> perm;
        if (ok) {
            MT.assign((size_t)nv * nv, 0.0);  // M^T, M = rows B of [A | -F ; 0 | A_t]
            for (int r = 0; r < nv; ++r) for (int j = 0; j < nv; ++j) MT[(size_t)j * nv + r] = Alp[(size_t)B[r] * nv + j];
            ok = lu_factor(MT, nv, perm);
        }
        if (ok) {
            std::vector<char> inB(rows_x, 0);
            for (int r : B) inB[r] = 1;
            std::vector<int> colpos;  // positions in B of the inequality rows (the alive nonbasic columns)
            for (int r = 0; r < nv; ++r) if (B[r] >= ne) { colpos.push_back(r); d0_cols.push_back(B[r]); }
            std::vector<double> y(nv);
            const int nc0 = (int)d0_cols.size();
            for (int i = 0; i < rows_x && ok; ++i) {
                if (inB[i]) continue;
                lu_solve_host(MT, perm, nv, &Alp[(size_t)i * nv], y.data());  // y = a_i M^-1
                double beta = blp[i];
                for (int r = 0; r < nv; ++r) beta -= y[r] * blp[B[r]];
                if (beta < -1e-7) ok = false;
                d0_rows.push_back(i);
                d0.push_back(beta);
                for (int q = 0; q < nc0; ++q) d0.push_back(-y[colpos[q]]);
            }
        }
        if (!ok) { d0.clear(); d0_rows.clear(); d0_cols.clear(); }
        if (std::getenv("MPC_DEBUG_CREATE"))
            std::fprintf(stderr, "[mpc] create: base vertex rc %d status %d, %d tight rows of n_x + n_t = %d -> pre-crashed dictionary %s\n", rc0, (int)lp_status,
                         (int)B.size(), nv, ok ? "built" : "NOT built (no register-resident fast path for this program)");
    }
    // ---- vertex of the parameter polytope {A_t theta <= b_t} (kernels2.hpp) ----------------------------------------
    std::vector<double> tv_theta, tv_minv, tv_rows, d0T;
    std::vector<int> tv_tight;
    {
        bool ok = ntc >= nt;
        std::vector<int32_t> tight(std::max(ntc, 1), 0);
        int32_t lp_status = -1;
        if (ok) {
            std::vector<uint8_t> eqf(ntc, 0);
            if (ntc == nt) {
                // Exactly n_theta rows (round 6): if they are independent (the LU below decides) their common point is the set's only vertex
                // -- a pointed cone, e.g. the lower bounds of theta that a presolve leaves when the main rows imply the upper ones
                // (profiles/r05_sweep.json: mpqp_10_2_20 / _30 ran on the LDS-engine kernels for want of this case).
                for (int i = 0; i < ntc; ++i) tight[i] = 1;
                lp_status = LP_OPTIMAL;
            } else {
                int rc0 = lp_batch_impl(device, 1, ntc, nt, p->A_t, 1, p->b_t, 1, nullptr, 1, eqf.data(), &lp_status, nullptr, nullptr, nullptr, tight.data());
                HIP_TRY(nullptr, hipSetDevice(device));
                ok = rc0 == MPC_OK && lp_status == LP_OPTIMAL;
            }
        }
        std::vector<int> B;
        for (int i = 0; i < ntc && ok; ++i) if (tight[i]) B.push_back(i);
        ok = ok && (int)B.size() == nt;
        std::vector<double> M, MTf;
        std::vector<int> perm;
        if (ok) {
            M.assign((size_t)nt * nt, 0.0);
            for (int r = 0; r < nt; ++r) for (int j = 0; j < nt; ++j) M[(size_t)r * nt + j] = p->A_t[(size_t)B[r] * nt + j];
            std::vector<double> LU(M);
            ok = lu_factor(LU, nt, perm);
            if (ok) {
                tv_minv.assign((size_t)nt * nt, 0.0);
                std::vector<double> e(nt), x(nt);
                for (int j = 0; j < nt; ++j) {   // column j of M^-1
                    std::fill(e.begin(), e.end(), 0.0); e[j] = 1.0;
                    lu_solve_host(LU, perm, nt, e.data(), x.data());
                    for (int t = 0; t < nt; ++t) tv_minv[(size_t)t * nt + j] = x[t];
                }
                std::vector<double> bB(nt);
                for (int r = 0; r < nt; ++r) bB[r] = p->b_t[B[r]];
                tv_theta.assign(nt, 0.0);
                lu_solve_host(LU, perm, nt, bB.data(), tv_theta.data());
                std::vector<char> inB(ntc, 0);
                for (int r : B) inB[r] = 1;
                for (int i = 0; i < ntc; ++i) {
                    if (inB[i]) continue;
                    double mx = 0; for (int t = 0; t < nt; ++t) mx = std::max(mx, std::fabs(p->A_t[(size_t)i * nt + t]));
                    double sc = 1.0;
                    if (mx > 1e-8) { int ex; std::frexp(mx, &ex); sc = std::ldexp(1.0, -ex); }
                    double beta = p->b_t[i] * sc;
                    for (int t = 0; t < nt; ++t) beta -= (mx > 1e-8 ? p->A_t[(size_t)i * nt + t] * sc : 0.0) * tv_theta[t];
                    tv_rows.push_back(beta);
                    for (int j = 0; j < nt; ++j) {
                        double acc = 0; for (int t = 0; t < nt; ++t) acc += (mx > 1e-8 ? p->A_t[(size_t)i * nt + t] * sc : 0.0) * tv_minv[(size_t)t * nt + j];
                        tv_rows.push_back(-acc);
                    }
                }
            }
        }
        if (!ok) { tv_theta.clear(); tv_minv.clear(); tv_rows.clear(); } else tv_tight = B;
        if (!d0.empty()) {
            const int mr = (int)d0_rows.size(), cc = 1 + (int)d0_cols.size();
            d0T.assign((size_t)mr * cc, 0.0);
            for (int i = 0; i < mr; ++i) for (int j = 0; j < cc; ++j) d0T[(size_t)j * mr + i] = d0[(size_t)i * cc + j];
        }
    }
    // ---- one device allocation for every read-only block -------------------------------------------------
    std::vector<double> host;
    auto put = [&](const double *src, size_t cnt) { size_t off = host.size(); host.insert(host.end(), src, src + cnt); while (host.size() % 2) host.push_back(0.0); return off; };
    std::vector<double> zeros((size_t)std::max(nx * nx, 1), 0.0);
    const size_t oA = put(p->A, (size_t)nc * nx), ob = put(p->b, nc), oF = put(p->F, (size_t)nc * nt), oc = put(p->c, nx), oH = put(p->H, (size_t)nx * nt);
    const size_t oQ = put(p->Q ? p->Q : zeros.data(), (size_t)nx * nx);
    const size_t oAt = put(ntc ? p->A_t : zeros.data(), (size_t)std::max(ntc * nt, 1)), obt = put(ntc ? p->b_t : zeros.data(), std::max(ntc, 1));
    // the blocks the set-up kernel fills are reserved at full size (zero until then); with MPC_HOST_SETUP=1 they are filled here
    auto put_or_reserve = [&](const std::vector<double> &v, size_t cnt) { std::vector<double> z; if (v.empty()) z.assign(std::max<size_t>(cnt, 1), 0.0); return put(v.empty() ? z.data() : v.data(), std::max<size_t>(cnt, 1)); };
    const bool dev_schur = p->Q && !host_setup;
    const size_t oW = put_or_reserve(W, (mode == 0 || dev_schur) ? (size_t)nc * nc : 1), oUV = put_or_reserve(UV, (mode == 0 || dev_schur) ? (size_t)nc * nr : 1);
    const size_t oGt = put_or_reserve(Gt, (mode == 0 || dev_schur) ? (size_t)nc * nx : 1), oX0H = put_or_reserve(X0H, (mode == 0 || dev_schur) ? (size_t)nx * nr : 1);
    std::vector<double> AAT;
    if (host_setup) {
        AAT.assign((size_t)nc * nc, 0.0);
        for (int i = 0; i < nc; ++i) for (int j = 0; j <= i; ++j) {
            double sdot = 0; for (int l = 0; l < nx; ++l) sdot += p->A[(size_t)i * nx + l] * p->A[(size_t)j * nx + l];
            AAT[(size_t)i * nc + j] = sdot; AAT[(size_t)j * nc + i] = sdot;
        }
    }
    const size_t oAAT = put_or_reserve(AAT, (size_t)nc * nc);
    std::vector<double> ATr((size_t)std::max(nx * nc, 1), 0.0);      // A transposed (k_region2's row build reads a column of A per step)
    for (int i = 0; i < nc; ++i) for (int l = 0; l < nx; ++l) ATr[(size_t)l * nc + i] = p->A[(size_t)i * nx + l];
    const size_t oATr = put(ATr.data(), ATr.size());
    // equality rows eliminated from the Schur blocks (setup_mfma.hpp): lets k_kkt_thread take active sets of ne + (1..8) rows
    const bool want_elim = ne > 0 && dev_schur && ![] { const char *ev = std::getenv("MPC_NO_EQ_ELIM"); return ev && ev[0] == '1'; }();
    const std::vector<double> none;
    const size_t oWr = put_or_reserve(none, want_elim ? (size_t)nc * nc : 1), oUVr = put_or_reserve(none, want_elim ? (size_t)nc * nr : 1);
    const size_t oAATr = put_or_reserve(none, want_elim ? (size_t)nc * nc : 1), oMe = put_or_reserve(none, want_elim ? (size_t)ne * nr : 1);
    const size_t oNe = put_or_reserve(none, want_elim ? (size_t)ne * nc : 1), ogE = put_or_reserve(none, want_elim ? (size_t)2 * ne : 1);
    std::vector<double> UVr;
    bool elim_ok = false;
    const size_t obase = put(base.data(), base.size());
    const size_t od0 = put(d0.empty() ? zeros.data() : d0.data(), d0.empty() ? 1 : d0.size());
    const size_t od0T = put(d0T.empty() ? zeros.data() : d0T.data(), d0T.empty() ? 1 : d0T.size());
    const size_t otvt = put(tv_theta.empty() ? zeros.data() : tv_theta.data(), tv_theta.empty() ? 1 : tv_theta.size());
    const size_t otvm = put(tv_minv.empty() ? zeros.data() : tv_minv.data(), tv_minv.empty() ? 1 : tv_minv.size());
    const size_t otvr = put(tv_rows.empty() ? zeros.data() : tv_rows.data(), tv_rows.empty() ? 1 : tv_rows.size());
    HIP_TRY(nullptr, h->blocks.ensure(host.size() * sizeof(double), h->stream));
    HIP_TRY(nullptr, hipMemcpyAsync(h->blocks.p, host.data(), host.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (!host_setup) {
        // ---- the dense Hessian factor and the one-off Schur blocks on the matrix cores (setup_mfma.hip) ----------------------
        double *db = h->blocks.as<double>();
        SetupJob job{};
        job.nx = nx; job.nt = nt; job.nc = nc; job.NP = setup_pad16(nx); job.MP = setup_pad16(nc); job.RP = setup_pad16(nr);
        job.A = db + oA; job.b = db + ob; job.F = db + oF; job.c = db + oc; job.H = db + oH; job.Q = p->Q ? db + oQ : nullptr;
        job.W = db + oW; job.UV = db + oUV; job.Gt = db + oGt; job.X0H = db + oX0H; job.AAT = db + oAAT;
        DevBuf work, jobbuf;
        const size_t job_bytes = (sizeof(SetupJob) + 15) & ~size_t(15);
        hipError_t e1 = work.ensure(setup_work_doubles(nx, nt, nc) * sizeof(double), h->stream);
        hipError_t e2 = e1 == hipSuccess ? jobbuf.ensure(job_bytes + 16, h->stream) : e1;
        int flag = 1, flag_e = 1;
        if (e2 == hipSuccess) {
            job.work = work.as<double>();
            job.flag = reinterpret_cast<int *>(jobbuf.as<char>() + job_bytes);
            if (want_elim) {
                job.ne = ne; job.Wr = db + oWr; job.UVr = db + oUVr; job.AATr = db + oAATr; job.Me = db + oMe; job.Ne = db + oNe; job.gE = db + ogE;
                job.flag_e = job.flag + 1;
            }
            e2 = hipMemcpyAsync(jobbuf.p, &job, sizeof(SetupJob), hipMemcpyHostToDevice, h->stream);
            if (e2 == hipSuccess) e2 = setup_launch(jobbuf.as<SetupJob>(), 1, h->stream);
            if (e2 == hipSuccess) e2 = hipMemcpyAsync(&flag, job.flag, sizeof(int), hipMemcpyDeviceToHost, h->stream);
            if (e2 == hipSuccess && p->Q) { UV.assign((size_t)nc * nr, 0.0); e2 = hipMemcpyAsync(UV.data(), job.UV, UV.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream); }
            if (e2 == hipSuccess && want_elim) {
                UVr.assign((size_t)nc * nr, 0.0);
                e2 = hipMemcpyAsync(UVr.data(), job.UVr, UVr.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
                if (e2 == hipSuccess) e2 = hipMemcpyAsync(&flag_e, job.flag_e, sizeof(int), hipMemcpyDeviceToHost, h->stream);
            }
            if (e2 == hipSuccess) e2 = hipStreamSynchronize(h->stream);
        }
        work.release(); jobbuf.release();   // the stream has been synchronised (or nothing was queued)
        if (e2 != hipSuccess) return fail(nullptr, MPC_ERR_HIP, std::string("MFMA set-up kernel: ") + hipGetErrorString(e2));
        mode = (p->Q && flag == 0) ? 0 : 1;
        elim_ok = want_elim && mode == 0 && flag_e == 0;
        h->elim_ok = elim_ok;
        h->o_elim[0] = oWr; h->o_elim[1] = oUVr; h->o_elim[2] = oAATr; h->o_elim[3] = oMe; h->o_elim[4] = oNe; h->o_elim[5] = ogE;
    }
    HIP_TRY(nullptr, hipStreamSynchronize(h->stream));
    h->kkt_mode = mode;
    const double *d = h->blocks.as<double>();
    DevProblem P{};
    P.n_x = nx; P.n_t = nt; P.n_c = nc; P.n_eq = ne; P.n_tc = ntc; P.is_qp = h->is_qp; P.kkt_mode = mode;
    P.A = d + oA; P.b = d + ob; P.F = d + oF; P.c = d + oc; P.H = d + oH; P.Q = d + oQ; P.A_t = d + oAt; P.b_t = d + obt;
    P.W = d + oW; P.UV = d + oUV; P.Gt = d + oGt; P.X0H = d + oX0H; P.base = d + obase; P.AAT = d + oAAT; P.AT = d + oATr;
    P.d0T = d + od0T; P.tv_theta = d + otvt; P.tv_minv = d + otvm; P.tv_rows = d + otvr;
    P.has_tv = tv_theta.empty() ? 0 : 1; P.n_tpre = tv_theta.empty() ? 0 : ntc - nt;
    P.d0 = d + od0; P.has_d0 = d0.empty() ? 0 : 1; P.n_d0r = (int)d0_rows.size(); P.n_d0c = (int)d0_cols.size();
    {
        std::vector<int> maps(d0_rows);
        maps.insert(maps.end(), d0_cols.begin(), d0_cols.end());
        maps.insert(maps.end(), tv_tight.begin(), tv_tight.end());
        if (maps.empty()) maps.push_back(0);
        HIP_TRY(nullptr, h->iblocks.ensure(maps.size() * sizeof(int), h->stream));
        HIP_TRY(nullptr, hipMemcpyAsync(h->iblocks.p, maps.data(), maps.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(nullptr, hipStreamSynchronize(h->stream));
        P.d0_rows = h->iblocks.as<int>();
        P.d0_cols = h->iblocks.as<int>() + d0_rows.size();
        P.tv_tight = h->iblocks.as<int>() + d0_rows.size() + d0_cols.size();
    }
    // ---- LDS layouts ---------------------------------------------------------------------------------------
    const int kmax = std::min(nc, nx);
    const int rows_t = nc - ne + ntc;
    P.kmax = kmax;
    P.ld_x = odd_at_least(nx + nt + 3);
    P.ld_t = odd_at_least(nt + 4);
    const int size_K = mode == 0 ? std::max(kmax * kmax + kmax, kmax * nx) : std::max({(nx + kmax) * (nx + kmax) + (nx + kmax) * nr, kmax * nx, nx * nx});
    const int size_L = std::max(kmax * nr, 1), size_X = nx * nr;
    const int T_v = std::max((rows_x + 1) * P.ld_x, (rows_t + 1) * P.ld_t);
    const int T_r = std::max((rows_t + 2) * P.ld_t, rows_t * nr);
    const Layout lv = make_layout(T_v, size_K, size_L, 0, mode == 1 ? size_X : 0, kmax, nc, std::max(P.ld_x, P.ld_t), rows_x + 1, rows_t);
    const Layout lr = make_layout(T_r, size_K, size_L, rows_t * nr, size_X, kmax, nc, P.ld_t, rows_t + 2, rows_t);
    h->Pv = P; apply_layout(h->Pv, lv); h->lds_v = lv.bytes;
    h->Pr = P; apply_layout(h->Pr, lr); h->lds_r = lr.bytes;
    if (h->lds_v > 160 * 1024 || h->lds_r > 160 * 1024) return fail(nullptr, MPC_ERR_INVALID, "problem does not fit the 160 KiB LDS of one CU");
    if (h->lds_v > 48 * 1024) HIP_TRY(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(k_verdict), hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_v));
    if (h->lds_r > 48 * 1024) {
        HIP_TRY(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(k_region<RG_FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_r));
        HIP_TRY(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(k_region<RG_FACET>), hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_r));
        HIP_TRY(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(k_region<RG_ASSEMBLE>), hipFuncAttributeMaxDynamicSharedMemorySize, h->lds_r));
    }
    h->grid_v = h->n_cu * waves_per_cu(h->lds_v);
    // fast path (k_verdict2): needs the theta vertex, the pre-crashed dictionary and sizes inside the instantiations
    bool box_done = false, box_open = false;   // the bounding box of the parameter set has been computed / has an infinite side
    {
        const int rows_th = rows_t - nt, rows_x = P.n_d0r;
        const int slots_t = rows_th <= 64 ? 1 : (rows_th <= 128 ? 2 : 0), slots_x = rows_x <= 64 ? 1 : (rows_x <= 128 ? 2 : 0);
        const int tsel = nt <= 4 ? 0 : (nt <= 8 ? 1 : (nt <= 10 ? 2 : -1));   // kernels are instantiated for n_theta <= 4, 8, 10
        const int xsel = P.n_d0c <= 14 ? 0 : (P.n_d0c <= 30 ? 1 : -1);
        if (std::getenv("MPC_DEBUG_CREATE"))
            std::fprintf(stderr, "[mpc] create: vertex of the parameter set %d, dictionary %d (rows %d, columns %d), theta rows %d -> slots %d / %d, n_theta class %d, column class %d: register-resident kernels %s\n",
                         (int)P.has_tv, (int)P.has_d0, rows_x, (int)P.n_d0c, rows_th, slots_t, slots_x, tsel, xsel,
                         (P.has_tv && P.has_d0 && slots_t && slots_x && tsel >= 0 && xsel >= 0) ? "ON" : "OFF (LDS-engine kernels)");
        if (P.has_tv && P.has_d0 && slots_t && slots_x && tsel >= 0 && xsel >= 0) {
            h->fast = 1;
            h->fast_t = tsel * 2 + (slots_t - 1);
            h->fast_x = xsel * 2 + (slots_x - 1);
            if (xsel >= 1) h->x2_wpc = std::min(h->x2_wpc, 4 * X2_WAVES);   // the 32-column instantiations of k_x2 are built for two wavefronts per SIMD
            // zero-padded blocks of k_theta2 (ThetaArgs) for its compile-time NT
            const int NTP = tsel == 0 ? 4 : (tsel == 1 ? 8 : 10), LS = NTP + 1;
            {
                std::vector<double> tb;
                const size_t oUVp = tb.size(); tb.resize(tb.size() + (size_t)nc * LS, 0.0);
                if (mode == 0) for (int i = 0; i < nc; ++i) for (int t = 0; t < nr; ++t) tb[oUVp + (size_t)i * LS + t] = UV[(size_t)i * nr + t];
                const size_t oUVrp = tb.size();
                if (elim_ok) {
                    tb.resize(tb.size() + (size_t)nc * LS, 0.0);
                    for (int i = 0; i < nc; ++i) for (int t = 0; t < nr; ++t) tb[oUVrp + (size_t)i * LS + t] = UVr[(size_t)i * nr + t];
                }
                const size_t otvp = tb.size(); tb.resize(tb.size() + (size_t)NTP * NTP + 3 * NTP, 0.0);
                for (int t = 0; t < nt; ++t) for (int j = 0; j < nt; ++j) tb[otvp + (size_t)t * NTP + j] = tv_minv[(size_t)t * nt + j];
                for (int t = 0; t < nt; ++t) tb[otvp + (size_t)NTP * NTP + t] = tv_theta[t];
                {   // bounding box of the parameter polytope: 2 n_t LPs on the device (an unbounded direction gives +-inf)
                    std::vector<double> cc((size_t)2 * nt * nt, 0.0), obj(2 * nt, 0.0);
                    for (int t = 0; t < nt; ++t) { cc[(size_t)(2 * t) * nt + t] = 1.0; cc[(size_t)(2 * t + 1) * nt + t] = -1.0; }
                    std::vector<uint8_t> eqz((size_t)2 * nt * ntc, 0);
                    std::vector<int32_t> stt(2 * nt, -1);
                    int rcb = lp_batch_impl(device, 2 * nt, ntc, nt, p->A_t, 1, p->b_t, 1, cc.data(), 0, eqz.data(), stt.data(), nullptr, obj.data(), nullptr, nullptr);
                    HIP_TRY(nullptr, hipSetDevice(device));
                    for (int t = 0; t < nt; ++t) {
                        const bool ok_lo = rcb == MPC_OK && stt[2 * t] == LP_OPTIMAL, ok_hi = rcb == MPC_OK && stt[2 * t + 1] == LP_OPTIMAL;
                        // the LP optimum carries the simplex tolerance: widen the box a little (the screen only needs an outer box)
                        const double lo = ok_lo ? obj[2 * t] : -INFINITY, hi = ok_hi ? -obj[2 * t + 1] : INFINITY;
                        box_done = true; box_open = box_open || !ok_lo || !ok_hi;
                        const double pad = 1e-6 * (1.0 + std::max(std::fabs(ok_lo ? lo : 0.0), std::fabs(ok_hi ? hi : 0.0)));
                        tb[otvp + (size_t)NTP * NTP + NTP + t] = lo - pad;
                        tb[otvp + (size_t)NTP * NTP + 2 * NTP + t] = hi + pad;
                    }
                }
                const int npre = ntc - nt;
                const size_t otvr = tb.size(); tb.resize(tb.size() + (size_t)std::max(npre, 1) * LS, 0.0);
                for (int i = 0; i < npre; ++i) for (int t = 0; t < nr; ++t) tb[otvr + (size_t)i * LS + t] = tv_rows[(size_t)i * nr + t];
                HIP_TRY(nullptr, h->theta_blocks.ensure(tb.size() * sizeof(double), h->stream));
                HIP_TRY(nullptr, hipMemcpyAsync(h->theta_blocks.p, tb.data(), tb.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(nullptr, hipStreamSynchronize(h->stream));
                const double *tbd = h->theta_blocks.as<double>();
                { const char *ev = std::getenv("MPC_KKT_BOX_SELECT"); h->targs.box_finite = (box_done && !box_open && !(ev && ev[0] == '1')) ? 1 : 0; }   // (=1: the select form of the screen's row test on every program; tests)
                h->targs.W = P.W; h->targs.UVp = tbd + oUVp; h->targs.tvp = tbd + otvp; h->targs.tv_rows = tbd + otvr; h->targs.chunk = 1;
                h->targs.ne = elim_ok ? ne : 0;
                {   // MPC_TH_DIV (work items per wavefront of k_theta2, default 4; 0 = round-3 behaviour) / MPC_TH_MAXW (waves per SIMD, default 2)
                    const char *e1 = std::getenv("MPC_TH_DIV"), *e2 = std::getenv("MPC_TH_MAXW");
                    h->targs.wave_div = e1 ? std::atoi(e1) : 4;
                    h->targs.wave_max = (e2 ? std::atoi(e2) : 2) * 4 * h->n_cu;
                    if (h->targs.wave_div <= 0) { h->targs.wave_div = 0; h->targs.wave_max = 0; }
                }
                h->targs.Wr = elim_ok ? d + oWr : P.W; h->targs.UVrp = elim_ok ? tbd + oUVrp : h->targs.UVp; h->targs.AATr = elim_ok ? d + oAATr : P.AAT;
                h->targs.Me = d + oMe; h->targs.Ne = d + oNe; h->targs.gE = d + ogE;
            }
            const Layout lf = make_layout(NTP * NTP + 3 * NTP + kmax * LS, size_K, size_L, 0, mode == 1 ? size_X : 0, kmax, nc, 2, 2, 2);
            h->Pf = P; apply_layout(h->Pf, lf); h->lds_f = lf.bytes;
            h->targs.cbuf_off = 0;
            if (slots_t == 2) {   // scratch of the live-row compaction of k_theta2<., 2>: 64 rows of NT+4 doubles and 2 ints
                h->targs.cbuf_off = h->lds_f / 8;
                h->lds_f += 64 * (NTP + 4) * 8 + 64 * 2 * 4;
                h->lds_f = (h->lds_f + 15) & ~15;
            }
            h->grid_f = h->n_cu * std::min(16, waves_per_cu(h->lds_f));
            HIP_TRY(nullptr, h->pf_dev.ensure(sizeof(DevProblem), h->stream));
            HIP_TRY(nullptr, hipMemcpyAsync(h->pf_dev.p, &h->Pf, sizeof(DevProblem), hipMemcpyHostToDevice, h->stream));
            HIP_TRY(nullptr, hipStreamSynchronize(h->stream));
            // k_region2: all rows_t region rows plus the cost row, n_t >= 2 (the one-parameter variant stays on k_region)
            const int slots_r = rows_t + 1 <= 64 ? 1 : (rows_t + 1 <= 128 ? 2 : 0);
            if (nt >= 2 && slots_r) {
                h->fast_r = tsel * 2 + (slots_r - 1);
                // (T: the staged [tv_minv | tv_theta | box] block, round 6)
                const Layout l2 = make_layout(NTP * NTP + 3 * NTP, std::max(size_K, nt * (2 * nt + 1)), size_L, rows_t * nr, size_X, kmax, nc, nt + 2, 2, rows_t);
                h->Pr2 = P; apply_layout(h->Pr2, l2); h->lds_r2 = l2.bytes;
                h->Pr2.tvp = h->targs.tvp;
                h->grid_r2 = h->n_cu * std::min(4 * (slots_r >= 2 ? 2 : R2W), waves_per_cu(h->lds_r2));   // waves per SIMD of the launch bounds
                HIP_TRY(nullptr, h->pr2_dev.ensure(sizeof(DevProblem), h->stream));
                HIP_TRY(nullptr, hipMemcpyAsync(h->pr2_dev.p, &h->Pr2, sizeof(DevProblem), hipMemcpyHostToDevice, h->stream));
                HIP_TRY(nullptr, hipStreamSynchronize(h->stream));
            }
        }
    }
    // ---- is the parameter set open in some direction?  (then the reference's optimality LP can be unbounded: k_recession) -------------
    {
        const char *ev = std::getenv("MPC_NO_RECESSION");
        bool open = false;
        if (!(ev && ev[0] == '1')) {
            if (nc == ne) open = true;                                     // no multiplier / slack row at all bounds t (the base set)
            else if (nt > 0 && (ntc == 0 || tv_theta.empty())) open = true;   // no row, or no vertex: the set contains a line (or the search failed: the test is always safe)
            else if (nt > 0) {
                if (!box_done) {   // a vertex, but maybe an open cone: the bounding box (2 n_t LPs on the device, as for the register-resident kernels)
                    std::vector<double> cc((size_t)2 * nt * nt, 0.0), obj(2 * nt, 0.0);
                    for (int t = 0; t < nt; ++t) { cc[(size_t)(2 * t) * nt + t] = 1.0; cc[(size_t)(2 * t + 1) * nt + t] = -1.0; }
                    std::vector<uint8_t> eqz((size_t)2 * nt * ntc, 0);
                    std::vector<int32_t> stt(2 * nt, -1);
                    const int rcb = lp_batch_impl(device, 2 * nt, ntc, nt, p->A_t, 1, p->b_t, 1, cc.data(), 0, eqz.data(), stt.data(), nullptr, obj.data(), nullptr, nullptr);
                    HIP_TRY(nullptr, hipSetDevice(device));
                    for (int t = 0; t < 2 * nt; ++t) box_open = box_open || rcb != MPC_OK || stt[t] != LP_OPTIMAL;
                }
                open = box_open;
            }
        }
        h->theta_open = open;
        if (std::getenv("MPC_DEBUG_CREATE")) std::fprintf(stderr, "[mpc] create: parameter set %s\n", open ? "OPEN in some direction (k_recession behind the verdict stages)" : "bounded");
    }
    h->grid_r = h->n_cu * waves_per_cu(h->lds_r);
    h->rec_d = (long long)nx * nt + nx + (long long)nc * nt + nc + (long long)(nc + ntc) * nt + (nc + ntc);
    h->rec_i = 5 + (long long)nc + ntc + nc + nc + nc;
    HIP_TRY(nullptr, h->ctr.ensure(sizeof(LevelCounters), h->stream));
    HIP_TRY(nullptr, h->scratch.ensure(256, h->stream));
    {
        void *hp = nullptr, *dp = nullptr;
        HIP_TRY(nullptr, host_pool_take(4096, &hp, nullptr, true));
        HIP_TRY(nullptr, hipHostGetDevicePointer(&dp, hp, 0));
        h->tot_host = static_cast<int32_t *>(hp);
        h->tot_dev = static_cast<int32_t *>(dp);
        std::memset(hp, 0, 64);
    }
    return MPC_OK;
}

extern "C" {

int mpc_destroy(mpc_handle *h) {
    if (!h) return MPC_OK;
    if (h->worker.joinable()) {
        { std::unique_lock<std::mutex> lk(h->wm); h->wcv.wait(lk, [&] { return !h->w_busy; }); h->w_req = 2; h->a_req.store(1, std::memory_order_release); }
        h->wcv.notify_all();
        h->worker.join();
    }
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    if (h->stream3) (void)hipStreamSynchronize(h->stream3);
    if (h->stream4) (void)hipStreamSynchronize(h->stream4);
    graveyard_flush(h);
    for (DevBuf *b : {&h->blocks, &h->iblocks, &h->frontier, &h->children, &h->status, &h->pruned, &h->flag, &h->pos, &h->opt_list,
                      &h->childmask, &h->count, &h->offset, &h->recd, &h->reci, &h->ctr, &h->pruned_b, &h->pruned_head, &h->scratch, &h->sums, &h->retry_list, &h->pf_dev, &h->pr2_dev, &h->headd, &h->headi, &h->epool, &h->facet_flags, &h->kkt_code, &h->kkt_L, &h->theta_blocks, &h->xq_groups, &h->xq_list, &h->x1_buf, &h->xretry_list, &h->theta_list, &h->vretry_list, &h->status_tmp, &h->part_counts, &h->part_lists, &h->kept_g, &h->done_g, &h->dict_d[0], &h->dict_d[1], &h->dict_i[0], &h->dict_i[1],
                      &h->dict_stored[0], &h->dict_stored[1], &h->parent_slot, &h->parent_slot_next, &h->dcnt}) b->release();
    if (h->tot_host) { (void)host_pool_give(h->tot_host); h->tot_host = h->tot_dev = nullptr; }
    for (HostBuf *b : {&h->st_list, &h->st_status, &h->st_hd, &h->st_hi, &h->st_pool, &h->st_fxd, &h->st_fxi, &h->st_rlist, &h->st_flags}) b->release();
    stream_release(h);
    solve_release(h);
    h->chunk_count.release();
    for (DevBuf *b : {&h->g.wave, &h->g.visited, &h->g.pending, &h->g.tmp_a, &h->g.tmp_b, &h->g.sort_tmp, &h->g.card, &h->g.idx, &h->g.card2, &h->g.idx2, &h->g.facet, &h->g.cnt, &h->g.off, &h->g.hist}) b->release();
    for (auto &e : h->ev) return_event(e, true);
    for (auto &e : h->kev) return_event(e, true);
    return_event(h->ev_hi, false);
    return_event(h->ev_fork, false);
    return_event(h->ev_join, false);
    return_event(h->ev_xfork, false);
    return_event(h->ev_part, false);
    return_event(h->ev_xjoin, false);
    return_event(h->ev_rfork, false);
    return_event(h->ev_rjoin, false);
    return_event(h->ev_rgo, false);
    return_stream(h->stream2);
    return_stream(h->stream3);
    return_stream(h->stream4);
    return_event(h->ev_rjoin2, false);
    return_event(h->ev_nth, false);
    return_event(h->ev_x1go, false);
    return_event(h->ev_x1done, false);
    if (h->own_stream) return_stream(h->stream);
    delete h;
    return MPC_OK;
}

// Gives the level buffers of an idle handle (frontier, children, statuses, lists, region records, both generations of the dictionary
// cache, the pruned list ...) back to the pool: the program's own blocks stay, a later level allocates again.  For a driver that holds
// many handles at once (mpc_level_run_batch) and wants the memory of those that are finished for the ones that are not.
int mpc_trim(mpc_handle *h) {
    if (!h) return MPC_ERR_INVALID;
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "a level started with mpc_level_start is still running"); }
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->stream2) HIP_TRY(h, hipStreamSynchronize(h->stream2));
    if (h->stream3) HIP_TRY(h, hipStreamSynchronize(h->stream3));
    if (h->stream4) HIP_TRY(h, hipStreamSynchronize(h->stream4));
    h->x1_pending = false; h->x1_stash = nullptr;
    graveyard_flush(h);
    stream_release(h);
    for (DevBuf *b : {&h->frontier, &h->children, &h->status, &h->pruned, &h->flag, &h->pos, &h->opt_list, &h->childmask, &h->count, &h->offset, &h->recd, &h->reci,
                      &h->sums, &h->retry_list, &h->headd, &h->headi, &h->epool, &h->facet_flags, &h->kkt_code, &h->kkt_L, &h->xq_groups, &h->xq_list, &h->x1_buf, &h->xretry_list, &h->theta_list, &h->vretry_list,
                      &h->status_tmp, &h->part_counts, &h->part_lists, &h->kept_g, &h->done_g, &h->dict_d[0], &h->dict_d[1], &h->dict_i[0], &h->dict_i[1],
                      &h->dict_stored[0], &h->dict_stored[1], &h->parent_slot, &h->parent_slot_next}) b->release();
    for (HostBuf *b : {&h->st_list, &h->st_status, &h->st_hd, &h->st_hi, &h->st_pool, &h->st_fxd, &h->st_fxi, &h->st_rlist, &h->st_flags}) b->release();
    h->n = 0; h->k = 0; h->n_pruned = 0; h->n_pruned_extra = 0; h->level_done = false;
    h->have_prev_dict = false; h->have_parent_slot = false;
    return MPC_OK;
}
// Device memory (GB) the next level of the handle's frontier will hold on the no-round-trip path (buffers sized by the number of
// candidates: region records, children, two generations of the dictionary cache): what mpc_level_batch_start charges against
// MPC_BATCH_BUDGET_GB.
double mpc_level_memory_gb(const mpc_handle *h, int32_t gen_children) { return h ? batch_level_gb(h, gen_children) : 0.0; }

int32_t mpc_mask_words(const mpc_handle *h) { return h ? h->mw : MPC_MASK_WORDS; }
int mpc_program_block(mpc_handle *h, int32_t which, double *out, int64_t cap, int64_t *n_out) {
    if (!h || !n_out) return MPC_ERR_INVALID;
    const long long nc = h->n_c, nx = h->n_x, nr = h->n_t + 1;
    const double *src = nullptr;
    long long n = 0;
    switch (which) {
        case 0: src = h->Pv.W; n = nc * nc; break;
        case 1: src = h->Pv.UV; n = nc * nr; break;
        case 2: src = h->Pv.Gt; n = nc * nx; break;
        case 3: src = h->Pv.X0H; n = nx * nr; break;
        case 4: src = h->Pv.AAT; n = nc * nc; break;
        // blocks with the equality rows eliminated (setup_mfma.hpp); empty when the program has none or the elimination is off
        case 5: src = h->blocks.as<double>() + h->o_elim[0]; n = nc * nc; break;
        case 6: src = h->blocks.as<double>() + h->o_elim[1]; n = nc * nr; break;
        case 7: src = h->blocks.as<double>() + h->o_elim[2]; n = nc * nc; break;
        case 8: src = h->blocks.as<double>() + h->o_elim[3]; n = (long long)h->n_eq * nr; break;
        case 9: src = h->blocks.as<double>() + h->o_elim[4]; n = (long long)h->n_eq * nc; break;
        case 10: src = h->blocks.as<double>() + h->o_elim[5]; n = 2LL * h->n_eq; break;
        default: return fail(h, MPC_ERR_INVALID, "mpc_program_block: which must be 0..10");
    }
    if (which < 4 && h->kkt_mode != 0) n = 0;   // the Schur blocks exist only for a positive definite Q
    if (which >= 5 && !h->elim_ok) n = 0;
    *n_out = n;
    if (n == 0) return MPC_OK;
    if (!out || cap < n) return fail(h, MPC_ERR_CAPACITY, "mpc_program_block: buffer too small");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(out, src, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MPC_OK;
}
int mpc_set_region_overlap(mpc_handle *h, int32_t on) { if (!h) return MPC_ERR_INVALID; h->no_roverlap = !on; return MPC_OK; }
int mpc_set_timing(mpc_handle *h, int32_t on) { if (!h) return MPC_ERR_INVALID; h->timing = on && !h->no_kev; return MPC_OK; }
int64_t mpc_region_doubles(const mpc_handle *h) { return h ? h->rec_d : 0; }
int64_t mpc_region_ints(const mpc_handle *h) { return h ? h->rec_i : 0; }
int32_t mpc_lds_bytes(const mpc_handle *h, int32_t which) { return !h ? 0 : (which == 0 ? h->lds_v : h->lds_r); }
void *mpc_stream(const mpc_handle *h) { return h ? reinterpret_cast<void *>(h->stream) : nullptr; }

// ---- frontier ----------------------------------------------------------------------------------------------
static int frontier_reset(mpc_handle *h, long long n, int k) {
    if (n < 0 || k < 0 || (n > 0 && k > h->n_c)) return fail(h, MPC_ERR_INVALID, "bad frontier shape");
    if (n > 0x7fffffffLL / std::max(k + 1, 1)) return fail(h, MPC_ERR_INVALID, "frontier too large for 32-bit offsets");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, h->frontier.ensure((size_t)std::max<long long>(n, 1) * std::max(k, 1) * sizeof(int32_t), h->stream));
    h->n = n; h->k = k; h->level_done = false; h->n_pruned_extra = 0; h->n_prev = 0; h->prev_regions = 0;
    h->have_prev_dict = false; h->have_parent_slot = false;   // a frontier set from outside has no cached parent dictionaries
    return MPC_OK;
}

int mpc_frontier_root(mpc_handle *h) {
    if (!h) return MPC_ERR_INVALID;
    const int cnt = h->n_c - h->n_eq;
    int rc = frontier_reset(h, cnt, h->n_eq + 1);
    if (rc) return rc;
    if (cnt > 0) hipLaunchKernelGGL(k_root_frontier, dim3((cnt + 63) / 64), dim3(64), 0, h->stream, h->n_eq, h->n_c, h->frontier.as<int32_t>());
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

int mpc_frontier_set(mpc_handle *h, const int32_t *cand, int64_t n, int32_t k) {
    if (!h || (!cand && n > 0 && k > 0)) return MPC_ERR_INVALID;
    int rc = frontier_reset(h, n, k);
    if (rc) return rc;
    if (n > 0 && k > 0) {
        HIP_TRY(h, hipMemcpyAsync(h->frontier.p, cand, (size_t)n * k * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return MPC_OK;
}

int mpc_frontier_set_device(mpc_handle *h, const int32_t *cand, int64_t n, int32_t k) {
    if (!h || (!cand && n > 0 && k > 0)) return MPC_ERR_INVALID;
    int rc = frontier_reset(h, n, k);
    if (rc) return rc;
    if (n > 0 && k > 0) {
        HIP_TRY(h, hipMemcpyAsync(h->frontier.p, cand, (size_t)n * k * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));  // caller-owned source
    }
    return MPC_OK;
}

int mpc_frontier_shard(mpc_handle *h, int32_t rank, int32_t world) {
    if (!h || world < 1 || rank < 0 || rank >= world) return MPC_ERR_INVALID;
    if (world == 1) return MPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const long long n = h->n, n_new = n > rank ? (n - rank + world - 1) / world : 0;
    const int k = h->k;
    hipStream_t st = h->stream;
    if (n_new > 0 && k > 0) {
        HIP_TRY(h, h->children.ensure((size_t)n_new * k * sizeof(int32_t), st));
        const long long tot = n_new * k;
        hipLaunchKernelGGL(k_take_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->frontier.as<int32_t>(), n_new, k, (long long)rank,
                           (long long)world, h->children.as<int32_t>());
        HIP_TRY(h, hipGetLastError());
        std::swap(h->frontier, h->children);
        if (h->have_parent_slot) {
            HIP_TRY(h, h->parent_slot_next.ensure((size_t)n_new * sizeof(int32_t), st));
            hipLaunchKernelGGL(k_take_rows, dim3((unsigned)((n_new + 255) / 256)), dim3(256), 0, st, h->parent_slot.as<int32_t>(), n_new, 1, (long long)rank,
                               (long long)world, h->parent_slot_next.as<int32_t>());
            HIP_TRY(h, hipGetLastError());
            std::swap(h->parent_slot, h->parent_slot_next);
        }
        HIP_TRY(h, hipStreamSynchronize(st));
    }
    h->n = n_new;
    h->n_prev = 0;   // h->children no longer holds the previous level's frontier (k_xq_thread's look-up of other parents is off for this level)
    h->level_done = false;
    return MPC_OK;
}

int mpc_frontier_info(const mpc_handle *h, int64_t *n, int32_t *k) {
    if (!h) return MPC_ERR_INVALID;
    if (n) *n = h->n;
    if (k) *k = h->k;
    return MPC_OK;
}

int mpc_frontier_get(mpc_handle *h, int32_t *cand, int64_t cap) {
    if (!h || !cand) return MPC_ERR_INVALID;
    if (cap < h->n) return fail(h, MPC_ERR_CAPACITY, "frontier buffer too small");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->n > 0 && h->k > 0) {
        HIP_TRY(h, hipMemcpyAsync(cand, h->frontier.p, (size_t)h->n * h->k * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return MPC_OK;
}

int mpc_pruned_clear(mpc_handle *h) { if (!h) return MPC_ERR_INVALID; h->n_pruned = 0; h->n_pruned_extra = 0; return MPC_OK; }

static int pruned_add(mpc_handle *h, const uint64_t *masks, int64_t m, hipMemcpyKind kind) {
    if (!h || (m > 0 && !masks) || m < 0) return MPC_ERR_INVALID;
    if (m == 0) return MPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    // A level that has run keeps its newly pruned masks right behind the list (they join it at mpc_frontier_advance): masks that
    // arrive in between -- the other ranks' -- go behind those.
    const long long tail = h->level_done ? h->n_pruned_new + h->n_pruned_extra : 0;
    HIP_TRY(h, h->pruned.ensure((size_t)(h->n_pruned + tail + m) * h->mw * sizeof(uint64_t), h->stream, true));
    HIP_TRY(h, hipMemcpyAsync(h->pruned.as<uint64_t>() + (size_t)(h->n_pruned + tail) * h->mw, masks, (size_t)m * h->mw * sizeof(uint64_t), kind, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // caller-owned source
    if (h->level_done) h->n_pruned_extra += m; else h->n_pruned += m;
    return MPC_OK;
}
int mpc_pruned_add(mpc_handle *h, const uint64_t *masks, int64_t m) { return pruned_add(h, masks, m, hipMemcpyHostToDevice); }
int mpc_pruned_add_device(mpc_handle *h, const uint64_t *masks, int64_t m) { return pruned_add(h, masks, m, hipMemcpyDeviceToDevice); }
int64_t mpc_pruned_count(const mpc_handle *h) { return h ? h->n_pruned : 0; }
int mpc_pruned_get(mpc_handle *h, uint64_t *masks, int64_t cap) {
    if (!h || !masks) return MPC_ERR_INVALID;
    if (cap < h->n_pruned) return fail(h, MPC_ERR_CAPACITY, "pruned buffer too small");
    if (h->n_pruned > 0) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(masks, h->pruned.p, (size_t)h->n_pruned * h->mw * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return MPC_OK;
}

// exclusive scan of n int32 values on the handle's stream (three launches); *total_dev receives the sum
static int launch_scan(mpc_handle *h, const int32_t *in, int32_t *out, long long n, int32_t *total_dev) {
    if (n <= SMALL_LEVEL_N) {   // one single-block launch
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SCAN_BLOCK), 0, h->stream, in, out, (int)n, total_dev);
        HIP_TRY(h, hipGetLastError());
        return MPC_OK;
    }
    const int nb = (int)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    HIP_TRY(h, h->sums.ensure((size_t)std::max(nb, 1) * sizeof(int32_t), h->stream));
    if (h->helper_it == 4) {   // four-wavefront workgroups (see block_exclusive_scan_it)
        hipLaunchKernelGGL(k_scan_block_sums<4>, dim3(nb), dim3(SCAN_BLOCK / 4), 0, h->stream, in, n, h->sums.as<int32_t>());
        hipLaunchKernelGGL(k_scan_sums<4>, dim3(1), dim3(SCAN_BLOCK / 4), 0, h->stream, h->sums.as<int32_t>(), nb, total_dev);
        hipLaunchKernelGGL(k_scan_apply<4>, dim3(nb), dim3(SCAN_BLOCK / 4), 0, h->stream, in, out, n, h->sums.as<int32_t>());
    } else {
        hipLaunchKernelGGL(k_scan_block_sums<1>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, in, n, h->sums.as<int32_t>());
        hipLaunchKernelGGL(k_scan_sums<1>, dim3(1), dim3(SCAN_BLOCK), 0, h->stream, h->sums.as<int32_t>(), nb, total_dev);
        hipLaunchKernelGGL(k_scan_apply<1>, dim3(nb), dim3(SCAN_BLOCK), 0, h->stream, in, out, n, h->sums.as<int32_t>());
    }
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

// LDS-engine region kernel over `list` (n_list optimal candidates), fixed-stride records at slots 0..n_list-1.
// Few candidates: one wavefront per (candidate, facet) and an assembly pass (latency form); many: one per candidate.
static int launch_region_v1(mpc_handle *h, const int32_t *list, long long n_list, int k, LevelCounters *ctr) {
    hipStream_t st = h->stream;
    HIP_TRY(h, h->recd.ensure((size_t)n_list * h->rec_d * sizeof(double), st));
    HIP_TRY(h, h->reci.ensure((size_t)n_list * h->rec_i * sizeof(int32_t), st));
    const long long rows_t = h->n_c - h->n_eq + h->n_tc;
    const bool split = h->n_t > 1 && rows_t > 0 && n_list <= 2LL * h->grid_r;
    HIP_TRY(h, hipMemsetAsync(&ctr->work_region, 0, sizeof(unsigned int), st));
    if (split) {
        HIP_TRY(h, h->facet_flags.ensure((size_t)n_list * rows_t, st));
        hipLaunchKernelGGL((k_region<RG_FACET>), dim3((unsigned)std::max<long long>(1, std::min<long long>(n_list * rows_t, 4LL * h->grid_r))), dim3(64), h->lds_r, st, h->Pr,   // (a program without inactive rows: rows_t = 0)
                           h->frontier.as<int32_t>(), k, list, (int)n_list, h->status.as<uint8_t>(), h->recd.as<double>(), h->reci.as<int32_t>(),
                           h->rec_d, h->rec_i, ctr, h->facet_flags.as<uint8_t>());
        HIP_TRY(h, hipMemsetAsync(&ctr->work_region, 0, sizeof(unsigned int), st));
        hipLaunchKernelGGL((k_region<RG_ASSEMBLE>), dim3((unsigned)std::min<long long>(n_list, h->grid_r)), dim3(64), h->lds_r, st, h->Pr,
                           h->frontier.as<int32_t>(), k, list, (int)n_list, h->status.as<uint8_t>(), h->recd.as<double>(), h->reci.as<int32_t>(),
                           h->rec_d, h->rec_i, ctr, h->facet_flags.as<uint8_t>());
    } else {
        hipLaunchKernelGGL((k_region<RG_FULL>), dim3((unsigned)std::min<long long>(n_list, h->grid_r)), dim3(64), h->lds_r, st, h->Pr,
                           h->frontier.as<int32_t>(), k, list, (int)n_list, h->status.as<uint8_t>(), h->recd.as<double>(), h->reci.as<int32_t>(),
                           h->rec_d, h->rec_i, ctr, (uint8_t *)nullptr);
    }
    HIP_TRY(h, hipGetLastError());
    return MPC_OK;
}

// ---- one level ------------------------------------------------------------------------------------------------
// tells a caller blocked in mpc_level_stream_info that the region stage of the running level has been launched (or that
// this level does not stream)
static void stream_ready(mpc_handle *h) {
    if (h->sv_cur >= 0) {
        // inside mpc_solve_start's loop: the region stage of level sv_cur has been launched -- its page-locked arrays are handed to
        // the caller of mpc_solve_level right away (a level that does not stream is published by the loop itself, after its fetch)
        auto &lv = h->sv_levels[h->sv_cur];
        if (h->so.active && !h->so.taken && !lv.ready.load(std::memory_order_relaxed)) {
            lv.mode = 1; lv.hd = h->so.hd; lv.hi = h->so.hi; lv.er = h->so.er;
            lv.n_slots = h->so.n_slots; lv.n_rows = h->so.cap_rows; lv.chunk = 1 << h->so.shift; lv.n_chunks = h->so.n_chunks;
            lv.flags = h->st_flags.as<int32_t>();
            h->so.taken = true;
            lv.ready.store(1, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(h->wm); }
            h->wcv.notify_all();
        }
        return;
    }
    std::lock_guard<std::mutex> lk(h->wm);
    h->w_stream_ready = true;
    h->a_ready.store(1, std::memory_order_release);
    h->wcv.notify_all();
}
static void stream_release(mpc_handle *h) {   // blocks of a streamed level nobody took
    // a level that ended early (error return) after launching its region kernel on stream3 may still be writing the blocks
    if (h->r3_dirty) { (void)hipStreamSynchronize(h->stream3); if (h->stream4) (void)hipStreamSynchronize(h->stream4); h->r3_dirty = false; }
    if (!h->so.taken) { if (h->so.hd) (void)host_pool_give(h->so.hd); if (h->so.hi) (void)host_pool_give(h->so.hi); if (h->so.er) (void)host_pool_give(h->so.er); }
    h->so = mpc_handle::StreamOut();
}


// ---- a level without host round trips ---------------------------------------------------------------------------------------
// The classic path below reads a list length back after almost every stage (to size the next launch): six synchronisations and
// ~25 dependent launches per level, 0.3-0.45 ms whatever the level computes -- the fixed cost of the first levels of every
// program, of all of config 2 and of the sub-programs of the mixed-integer enumeration.  Here every list length stays in device
// memory (h->dcnt; the kernels read it there: ThetaArgs::n_dev, DictCache::n_*_dev, RegionStream::n_opt_dev, k_verdict's n_dev),
// launches are sized by the one bound the host knows -- the number of candidates of the level --, buffers by the same bound,
// and the host synchronises ONCE, at the end.  Same kernels, same lists (the single-block compaction / partition kernels keep
// the candidate order), same decisions.  The rare stages whose buffers cannot be bounded cheaply -- the LDS-engine region kernel
// for candidates k_region2 gives up on -- are not part of it: if the level turns out to need them, it is repeated on the
// classic path (deterministic, so the repeat computes the same thing).
// dcnt layout (int32): [0] theta list | [12] doubtful after the theta stage | [4..7] classes after their re-solve: -, feasible,
// optimal (= the region launch), open | [8] open after the quick test | [24] doubtful after the (x,theta) stage | [17] optimal
// candidates that missed the region launch | [20] children
// Largest number of inequality rows k_kkt_thread solves for (one thread per candidate, everything in registers).  Round 6: 8 -> 10 -- the
// K = 9, 10 instantiations keep their k x k system and the k x (n_theta + 1) multipliers in 234-256 VGPRs (+ up to 108 AGPRs as spill space,
// one wavefront per SIMD at n_theta > 4) and still beat the wavefront-wide LDS solve inside k_theta2 several times over; K = 12
// needs 364 registers at every n_theta.  The shared launches of several programs (batch_level.hip) keep 8.
constexpr int KKT_THREAD_MAX = 10;
constexpr int KKT_SPREAD = BATCH_KKT_SPREAD, KKT_SPREAD_KMAX = BATCH_KKT_SPREAD_KMAX;   // small levels: lanes per candidate of k_kkt_thread, up to this many inequality rows
// every stream of the handle that may read or write dictionary records waits for the deferred k_x1 of the previous level (stream waits: the host does not block)
// issues a stashed k_x1 launch (MPC_X1_DEFER=2: behind the level's children stage, so that it runs beside the NEXT level's KKT kernel -- issue
// bound, no traffic -- instead of beside the children stage, which like k_x1 is bound by the memory system: config 4's k_children_count_b
// took 0.34 ms beside it, 0.08 alone)
static int x1_flush(mpc_handle *h) {
    if (!h->x1_stash) return MPC_OK;
    std::function<int()> f;
    f.swap(h->x1_stash);
    return f();
}
static int x1_join(mpc_handle *h) {
    { int rcf = x1_flush(h); if (rcf) return rcf; }
    if (!h->x1_pending) return MPC_OK;
    for (hipStream_t sx : {h->stream, h->stream2, h->stream3}) if (sx) HIP_TRY(h, hipStreamWaitEvent(sx, h->ev_x1done, 0));
    h->x1_pending = false;
    return MPC_OK;
}
static bool small_path_ok(const mpc_handle *h, long long n, int k, int32_t flags, int32_t gen_children) {
    if (h->no_smallpath || !h->fast || h->force_v1 || h->fast_r < 0 || n < 1 || n > h->smallpath_max) return false;
    if (flags & MPC_LEVEL_GRAPH) return false;
    if (h->test_late > 0 || h->debug_cycles) return false;
    // children and their parent slots are sized by the bound n (n_c - k) candidates of k + 1 indices
    const double child_bytes = (double)n * std::max(h->n_c - k, 1) * (k + 2) * 4.0;
    if (gen_children && child_bytes > 512e6) return false;
    const long long rows_t = h->n_c - h->n_eq + h->n_tc;
    const double rec_bytes = (double)n * ((double)(h->n_x * h->n_t + h->n_x + k * h->n_t + k) * 8.0 + (double)(8 + 2 * k + h->n_tc + 2 * (h->n_c - k)) * 4.0 +
                                          (double)rows_t * (h->n_t + 1) * 8.0);
    return rec_bytes <= 2e9;
}

// MPC_DEBUG_SMALL=1: at exit, how many no-round-trip levels had doubtful candidates (re-solved in place by the LDS engine) in their two verdict stages
static std::atomic<long long> g_small_levels{0}, g_small_retry_theta{0}, g_small_retry_x{0}, g_small_retry_cands{0}, g_small_repeats{0}, g_small_repeats_doubtful{0};
static void small_debug_note(const int32_t *cnt_host) {
    static const bool on = [] {
        const char *ev = std::getenv("MPC_DEBUG_SMALL");
        if (!(ev && ev[0] == '1')) return false;
        std::atexit([] { std::fprintf(stderr, "[mpc] small levels %lld: with doubtful candidates after the theta stage %lld, after the (x,theta) stage %lld (candidates %lld); repeated on the classic path %lld (because of doubtful candidates %lld)\n",
                                      g_small_levels.load(), g_small_retry_theta.load(), g_small_retry_x.load(), g_small_retry_cands.load(), g_small_repeats.load(), g_small_repeats_doubtful.load()); });
        return true;
    }();
    if (!on) return;
    g_small_levels++; g_small_retry_theta += (cnt_host[12] + cnt_host[4]) > 0; g_small_retry_x += cnt_host[24] > 0; g_small_retry_cands += cnt_host[12] + cnt_host[4] + cnt_host[24];
}

static int level_run_small(mpc_handle *h, int32_t gen_children, int32_t flags, mpc_level_stats *stats, bool *fallback) {
    *fallback = false;
    { int rcj = x1_join(h); if (rcj) return rcj; }
    const long long n = h->n;
    const int k = h->k;
    const size_t nn = (size_t)n;
    hipStream_t st = h->stream;
    h->n_opt = h->n_children = h->n_pruned_new = h->n_regions = 0;
    h->n_needx = 0;
    HIP_TRY(h, h->status.ensure(nn, st));
    HIP_TRY(h, h->pruned.ensure((size_t)(h->n_pruned + n) * h->mw * sizeof(uint64_t), st, true));   // the level's newly pruned masks go behind the list
    HIP_TRY(h, h->retry_list.ensure(nn * sizeof(int32_t), st));
    HIP_TRY(h, h->theta_list.ensure(nn * sizeof(int32_t), st));
    HIP_TRY(h, h->part_lists.ensure((size_t)PART_CLASSES * nn * sizeof(int32_t), st));
    HIP_TRY(h, h->dcnt.ensure(32 * sizeof(int32_t), st));
    {
        // counters, list lengths, the region kernel's completion flags and the "dictionary stored" flags: cleared by one launch
        const int nxc_ = h->fast_x >= 2 ? 32 : 16;
        const double need_gb_ = (double)nn * ((double)nxc_ * h->Pf.n_d0r * 8.0 + (double)dict_ints(h->Pf.n_d0r, nxc_, h->n_c) * 4.0) / 1e9;
        const bool will_store = gen_children && need_gb_ <= h->dict_budget_gb;
        HIP_TRY(h, h->done_g.ensure(nn * 2 * sizeof(unsigned int), st));
        if (will_store) HIP_TRY(h, h->dict_stored[h->dict_cur].ensure(nn, st));
        ZeroBufs z{};
        z.p[0] = h->ctr.p; z.bytes[0] = sizeof(LevelCounters);
        z.p[1] = h->dcnt.p; z.bytes[1] = 32 * sizeof(int32_t);
        z.p[2] = h->done_g.p; z.bytes[2] = nn * 2 * sizeof(unsigned int);
        if (will_store) { z.p[3] = h->dict_stored[h->dict_cur].p; z.bytes[3] = nn; }
        hipLaunchKernelGGL(k_zero_bufs, dim3((unsigned)std::min<size_t>(64, (nn * 8 + 2047) / 2048 + 1)), dim3(256), 0, st, z);
    }
    LevelCounters *ctr = h->ctr.as<LevelCounters>();
    int32_t *dcnt = h->dcnt.as<int32_t>();
    const int32_t *fr = h->frontier.as<int32_t>();
    uint8_t *stp = h->status.as<uint8_t>();
    const DevProblem *pf = h->pf_dev.as<DevProblem>();
    auto part_list = [&](int c) -> int32_t * { return h->part_lists.as<int32_t>() + (size_t)c * nn; };
    auto spec_of = [](std::initializer_list<std::pair<int, int>> classes) {
        unsigned long long spec = ~0ull;
        for (const auto &sc : classes) spec = (spec & ~(15ull << (4 * sc.first))) | ((unsigned long long)sc.second << (4 * sc.first));
        return spec;
    };
    const int blocks256 = (int)((n + 255) / 256);
    h->used_region2 = false; h->n_rretry = 0; h->n_erows = 0; h->rretry_rows = -1;
    h->fd = h->n_x * h->n_t + h->n_x + k * h->n_t + k;
    h->fi = 8 + k + h->n_tc + k + 2 * (h->n_c - k);
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[0], st));
    // ---- KKT solves + box screen, theta stage ----------------------------------------------------------------------------------
    const uint8_t *kkc = nullptr;
    const double *kkl = nullptr;
    const int32_t *theta_list = nullptr;
    ThetaArgs ta = h->targs;
    ta.chunk = 1;
    const int kd = k - h->targs.ne;   // rows the one-thread KKT kernel solves for (the equality rows are eliminated)
    if (h->kkt_mode == 0 && kd >= 1 && kd <= KKT_THREAD_MAX && !h->no_kkt_thread) {
        HIP_TRY(h, h->kkt_code.ensure(nn, st));
        HIP_TRY(h, h->kkt_L.ensure(nn * (size_t)k * (h->n_t + 1) * sizeof(double), st));
        kkc = h->kkt_code.as<uint8_t>(); kkl = h->kkt_L.as<double>();
        ThetaArgs tk = h->targs;
        // the kernel lists the candidates it leaves to the theta stage itself (one atomic per workgroup; a work list: its order changes nothing)
        const bool kkt_lists = !h->no_kkt_lists;
        if (kkt_lists) {
            HIP_TRY(h, h->xq_list.ensure(nn * sizeof(int32_t), st));
            tk.kt_list = h->theta_list.as<int32_t>(); tk.kt_n = dcnt + 0;
            tk.kx_list = h->xq_list.as<int32_t>(); tk.kx_n = dcnt + 10;
        }
        // KKT_SPREAD lanes per candidate while the active set is small (the kernel's comment): the level's 50 us floor becomes ~15
        const bool spread = h->kkt_spread > 0 && kd <= KKT_SPREAD_KMAX;
        const dim3 g((unsigned)(spread ? (n * KKT_SPREAD + 255) / 256 : blocks256)), b(256);
#define MPC_LAUNCH_KKT_(K_, SP_) if (h->fast_t >= 4) hipLaunchKernelGGL((k_kkt_thread<K_, 10, SP_>), g, b, 0, st, pf, fr, n, h->kkt_code.as<uint8_t>(), h->kkt_L.as<double>(), stp, tk, ctr); \
                                    else if (h->fast_t >= 2) hipLaunchKernelGGL((k_kkt_thread<K_, 8, SP_>), g, b, 0, st, pf, fr, n, h->kkt_code.as<uint8_t>(), h->kkt_L.as<double>(), stp, tk, ctr); \
                                    else hipLaunchKernelGGL((k_kkt_thread<K_, 4, SP_>), g, b, 0, st, pf, fr, n, h->kkt_code.as<uint8_t>(), h->kkt_L.as<double>(), stp, tk, ctr)
#define MPC_LAUNCH_KKT(K_) case K_: MPC_LAUNCH_KKT_(K_, 1); break
#define MPC_LAUNCH_KKT_S(K_) case K_: if (spread) { MPC_LAUNCH_KKT_(K_, KKT_SPREAD); } else { MPC_LAUNCH_KKT_(K_, 1); } break
        switch (kd) { MPC_LAUNCH_KKT_S(1); MPC_LAUNCH_KKT_S(2); MPC_LAUNCH_KKT_S(3); MPC_LAUNCH_KKT_S(4); MPC_LAUNCH_KKT_S(5); MPC_LAUNCH_KKT_S(6); MPC_LAUNCH_KKT(7); MPC_LAUNCH_KKT(8); MPC_LAUNCH_KKT(9); MPC_LAUNCH_KKT(10); }
#undef MPC_LAUNCH_KKT_S
#undef MPC_LAUNCH_KKT
#undef MPC_LAUNCH_KKT_
        if (!kkt_lists) hipLaunchKernelGGL(k_compact_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, ST_TODO, ST_TODO, h->theta_list.as<int32_t>(), dcnt + 0);
        theta_list = h->theta_list.as<int32_t>();
        ta.n_dev = dcnt + 0;
    }
    {
        const dim3 g((unsigned)std::min<long long>(n, h->grid_f)), b(64);
        switch (h->fast_t) {
            case 0: hipLaunchKernelGGL((k_theta2<4, 1>), g, b, h->lds_f, st, pf, fr, n, k, stp, ctr, kkc, kkl, ta, theta_list); break;
            case 1: hipLaunchKernelGGL((k_theta2<4, 2>), g, b, h->lds_f, st, pf, fr, n, k, stp, ctr, kkc, kkl, ta, theta_list); break;
            case 2: hipLaunchKernelGGL((k_theta2<8, 1>), g, b, h->lds_f, st, pf, fr, n, k, stp, ctr, kkc, kkl, ta, theta_list); break;
            case 3: hipLaunchKernelGGL((k_theta2<8, 2>), g, b, h->lds_f, st, pf, fr, n, k, stp, ctr, kkc, kkl, ta, theta_list); break;
            case 4: hipLaunchKernelGGL((k_theta2<10, 1>), g, b, h->lds_f, st, pf, fr, n, k, stp, ctr, kkc, kkl, ta, theta_list); break;
            default: hipLaunchKernelGGL((k_theta2<10, 2>), g, b, h->lds_f, st, pf, fr, n, k, stp, ctr, kkc, kkl, ta, theta_list); break;
        }
    }
    // the doubtful candidates of the theta stage are re-solved in place by the LDS engine first (the classic path does this on a
    // side stream under the (x,theta) stage), so that the partition below already knows every optimal candidate they yield
    // Round 5 (`fused`): doubtful candidates are rare (numerically doubtful pivots of the register simplex), so the level does not spend
    // three launches on them at each of its two verdict stages: ONE partition lists them as class 0 beside the classes below, the level
    // goes on without them, and if there was one (counted in [4] / [24]) the host repeats the level on the classic path -- as it does for
    // the other rare cases.  The end of the level is one launch (k_small_end) instead of five, the scan carries the publish.
    const bool fused = !h->theta_open && !h->no_small_fuse;
    if (!fused) {
        hipLaunchKernelGGL(k_partition_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, spec_of({{ST_RETRY, 0}}), h->part_lists.as<int32_t>(), (long long)n, dcnt + 12);
        hipLaunchKernelGGL(k_verdict, dim3((unsigned)std::min<long long>(n, 128)), dim3(64), h->lds_v, st, h->Pv, fr, n, k, stp, ctr, part_list(0), dcnt + 12);
        // open parameter set: "optimal" only if the reference's max-t LP is bounded (k_recession), decided before the region launch below
        if (h->theta_open) hipLaunchKernelGGL(k_recession, dim3((unsigned)std::min<long long>(n, 256)), dim3(64), h->lds_v, st, h->Pv, fr, n, k, stp);
    }
    // classes after the theta stage: [1] feasible, [2] optimal, [3] feasibility open ([0] doubtful, fused form only)
    hipLaunchKernelGGL(k_partition_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n,
                       fused ? spec_of({{ST_RETRY, 0}, {ST_FEASIBLE, 1}, {ST_OPT_PENDING, 2}, {ST_NEEDX, 3}, {ST_NEEDX_SING, 3}})
                             : spec_of({{ST_FEASIBLE, 1}, {ST_OPT_PENDING, 2}, {ST_NEEDX, 3}, {ST_NEEDX_SING, 3}}), h->part_lists.as<int32_t>(), (long long)n, dcnt + 4);
    HIP_TRY(h, hipGetLastError());
    // ---- region stage: one slot per optimal candidate, buffers sized by the bound.  It needs the theta stage's verdicts only; a
    // candidate that turns out optimal only later -- a doubtful one of the (x,theta) stage, re-solved -- sends the level to the classic
    // path.  (Measured: on its own stream beside the (x,theta) stage it gains nothing at this size -- the fork / join events cost what
    // the overlap of two 50-100 us kernels saves: config 2 1.76 ms against 1.68 in line; round 4, config 4, whose regions take 150 us:
    // 5.47 / 5.61 ms forked against 5.48 / 5.54 in line -- the region kernel itself is 0.185 instead of 0.158 ms beside k_x2.) -----------------------------------------------
    h->opt_ptr = part_list(2);
    SmallRX rx{};
    unsigned grid_r = 1;
    {
        const int rows_t_ = h->n_c - h->n_eq + h->n_tc;
        HIP_TRY(h, h->headd.ensure(nn * h->fd * sizeof(double), st));
        HIP_TRY(h, h->headi.ensure(nn * h->fi * sizeof(int32_t), st));
        HIP_TRY(h, h->epool.ensure(nn * rows_t_ * (h->n_t + 1) * sizeof(double), st));
        const int ldk = (rows_t_ + 1 + 63) & ~63;
        HIP_TRY(h, h->kept_g.ensure(nn * ldk, st));
        HIP_TRY(h, h->done_g.ensure(nn * 2 * sizeof(unsigned int), st));
        RegionStream rs{};
        rs.n_opt_dev = dcnt + 6;
        rs.w_cap = h->grid_r2; rs.w_max = h->rsplit_max;
        const int W = h->no_rsplit ? 1 : 0;   // 0: chosen in the kernel from the number of optimal candidates
        grid_r = (unsigned)std::min<long long>(n * std::max(h->rsplit_max, 1), h->grid_r2);
        const DevProblem *pr = h->pr2_dev.as<DevProblem>();
        const int nt_r = h->fast_r <= 1 ? 4 : (h->fast_r <= 3 ? 8 : 10);
        rx.pr = pr; rx.fr = h->frontier.as<int32_t>(); rx.k = k; rx.opt_list = h->opt_ptr; rx.n = (int)n; rx.status = h->status.as<uint8_t>();
        rx.headd = h->headd.as<double>(); rx.headi = h->headi.as<int32_t>(); rx.fd = h->fd; rx.fi = h->fi; rx.epool = h->epool.as<double>(); rx.ctr = ctr;
        rx.kkc = kkc; rx.kkl = kkl; rx.W = W; rx.kept_g = h->kept_g.as<uint8_t>(); rx.ldk = ldk; rx.done_g = h->done_g.as<unsigned int>();
        rx.tvp_box = h->no_rbox ? (const double *)nullptr : h->targs.tvp + (size_t)nt_r * nt_r + nt_r; rx.rs = rs;
        h->used_region2 = true;
    }
    // (the launch itself comes below: together with the (x,theta) kernel in one grid where an instantiation of the pair exists)
    auto launch_region2 = [&]() -> int {
        const dim3 g(grid_r), b(64);
#define MPC_LAUNCH_R2(NT_, SL_) hipLaunchKernelGGL((k_region2<NT_, SL_>), g, b, h->lds_r2, st, rx.pr, rx.fr, rx.k, rx.opt_list, rx.n, rx.status, rx.headd, rx.headi, rx.fd, rx.fi, rx.epool, \
                                                   rx.ctr, rx.kkc, rx.kkl, rx.W, rx.kept_g, rx.ldk, rx.done_g, rx.tvp_box, rx.rs)
        switch (h->fast_r) {
            case 0: MPC_LAUNCH_R2(4, 1); break;
            case 1: MPC_LAUNCH_R2(4, 2); break;
            case 2: MPC_LAUNCH_R2(8, 1); break;
            case 3: MPC_LAUNCH_R2(8, 2); break;
            case 4: MPC_LAUNCH_R2(10, 1); break;
            default: MPC_LAUNCH_R2(10, 2); break;
        }
#undef MPC_LAUNCH_R2
        HIP_TRY(h, hipGetLastError());
        return MPC_OK;
    };
    if (!fused) HIP_TRY(h, hipMemsetAsync(&ctr->work_retry, 0, sizeof(unsigned int), st));
    // ---- (x,theta) stage with the dictionary cache ---------------------------------------------------------------------------------
    const int nxc = h->fast_x >= 2 ? 32 : 16;
    h->dict_stride_d = (long long)nxc * h->Pf.n_d0r;
    h->dict_stride_i = dict_ints(h->Pf.n_d0r, nxc, h->n_c);
    DictCache dc{};
    dc.fresh_limit = h->x_fresh_limit; dc.second_max = h->x_second_max;
    dc.stride_d = h->dict_stride_d; dc.stride_i = h->dict_stride_i;
    if (h->have_prev_dict && h->have_parent_slot) {
        dc.parent_slot = h->parent_slot.as<int32_t>();
        dc.prev_d = h->dict_d[1 - h->dict_cur].as<double>();
        dc.prev_i = h->dict_i[1 - h->dict_cur].as<int32_t>();
    }
    h->storing = false;
    const double need_gb = (double)nn * (h->dict_stride_d * 8.0 + h->dict_stride_i * 4.0) / 1e9;
    if (gen_children && need_gb <= h->dict_budget_gb) {
        HIP_TRY(h, h->dict_d[h->dict_cur].ensure(nn * h->dict_stride_d * sizeof(double), st));
        HIP_TRY(h, h->dict_i[h->dict_cur].ensure(nn * h->dict_stride_i * sizeof(int32_t), st));
        HIP_TRY(h, h->dict_stored[h->dict_cur].ensure(nn, st));
        dc.cur_d = h->dict_d[h->dict_cur].as<double>(); dc.cur_i = h->dict_i[h->dict_cur].as<int32_t>();
        dc.stored = h->dict_stored[h->dict_cur].as<uint8_t>();
        h->storing = true;
        dc.pre1 = part_list(1); dc.n_pre1_dev = dcnt + 5;
        dc.pre2 = part_list(2); dc.n_pre2_dev = dcnt + 6;
    }
    dc.chunk = 1;
    const int32_t *needx_list = part_list(3);
    const int32_t *needx_n = dcnt + 7;
    const bool quick_test = !h->storing && dc.parent_slot && !h->no_xquick;
    bool merged_rx = false;
    if (fused && !quick_test && !h->no_small_rx) {
        // region kernel + (x,theta) kernel as ONE grid (batch_level.hpp, SmallRX): the two work on disjoint candidates
        DictCache d = dc;
        d.n_list_dev = needx_n;
        rx.pf = pf; rx.list = needx_list; rx.dc = d;
        hipError_t e_rx = hipSuccess;
        merged_rx = small_region2_x2_launch(h->fast_r, h->fast_x, grid_r, (unsigned)std::min<long long>(n, (long long)h->n_cu * 16), h->lds_r2, st, rx, &e_rx);
        if (merged_rx) HIP_TRY(h, e_rx);
    }
    if (!merged_rx) { int rcr = launch_region2(); if (rcr) return rcr; }
    if (quick_test) {   // last level: decisions only -- the quick test on a few vectors of the parent's dictionary first
        DictCache dq = dc;
        dq.n_list_dev = needx_n;
        const dim3 gg((unsigned)std::min<long long>(n, (long long)h->n_cu * 32)), bb(64);
        if (h->fast_x & 1) hipLaunchKernelGGL((k_xq<2>), gg, bb, 0, st, pf, fr, k, needx_list, (int)n, stp, ctr, dq, nxc);
        else hipLaunchKernelGGL((k_xq<1>), gg, bb, 0, st, pf, fr, k, needx_list, (int)n, stp, ctr, dq, nxc);
        hipLaunchKernelGGL(k_compact_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, ST_NEEDX, ST_NEEDX_SING, h->retry_list.as<int32_t>(), dcnt + 8);
        needx_list = h->retry_list.as<int32_t>();
        needx_n = dcnt + 8;
    }
    if (!merged_rx) {
        DictCache d = dc;
        d.n_list_dev = needx_n;
        // bound of the work items: every candidate once (open, or decided and expanded for its dictionary)
        const dim3 gg((unsigned)std::min<long long>(n, (long long)h->n_cu * 16)), bb(64);
        switch (h->fast_x) {
            case 0: hipLaunchKernelGGL((k_x2<16, 1>), gg, bb, 0, st, pf, fr, k, needx_list, (int)n, stp, ctr, d); break;
            case 1: hipLaunchKernelGGL((k_x2<16, 2>), gg, bb, 0, st, pf, fr, k, needx_list, (int)n, stp, ctr, d); break;
            case 2: hipLaunchKernelGGL((k_x2<32, 1>), gg, bb, 0, st, pf, fr, k, needx_list, (int)n, stp, ctr, d); break;
            default: hipLaunchKernelGGL((k_x2<32, 2>), gg, bb, 0, st, pf, fr, k, needx_list, (int)n, stp, ctr, d); break;
        }
    }
    HIP_TRY(h, hipGetLastError());
    const int keep_lowdim = (flags & MPC_LEVEL_KEEP_LOWDIM) ? 1 : 0;
    static_assert(sizeof(LevelCounters) + 64 + 32 * 4 <= 4096, "LevelCounters + list lengths must fit the pinned block");
    unsigned int *pub_dst = reinterpret_cast<unsigned int *>(h->tot_dev + 16);
    if (fused) {
        // doubtful candidates of the (x,theta) stage -> [24], optimal ones that missed the region launch -> [17], status histogram,
        // pruned masks (and, on a level without children, the publish): one launch
        if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev[1], st)); HIP_TRY(h, hipEventRecord(h->ev[2], st)); }
        unsigned long long *pout = h->pruned.as<unsigned long long>() + (size_t)h->n_pruned * h->mw;
        if (h->mw == 2) hipLaunchKernelGGL(k_small_end<2>, dim3(1), dim3(1024), 0, st, fr, (int)n, k, stp, pout, ctr, keep_lowdim, dcnt + 24, dcnt + 17,
                                           reinterpret_cast<const unsigned int *>(dcnt), 32, gen_children ? (unsigned int *)nullptr : pub_dst);
        else hipLaunchKernelGGL(k_small_end<4>, dim3(1), dim3(1024), 0, st, fr, (int)n, k, stp, pout, ctr, keep_lowdim, dcnt + 24, dcnt + 17,
                                reinterpret_cast<const unsigned int *>(dcnt), 32, gen_children ? (unsigned int *)nullptr : pub_dst);
        HIP_TRY(h, hipGetLastError());
    } else {
    // doubtful candidates of the (x,theta) stage (rare): re-solved in place as well
    hipLaunchKernelGGL(k_partition_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, spec_of({{ST_RETRY, 0}}), h->part_lists.as<int32_t>(), (long long)n, dcnt + 24);
    hipLaunchKernelGGL(k_verdict, dim3((unsigned)std::min<long long>(n, 128)), dim3(64), h->lds_v, st, h->Pv, fr, n, k, stp, ctr, part_list(0), dcnt + 24);
    HIP_TRY(h, hipGetLastError());
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], st));
    // a candidate that is still "optimal, region pending" now was not in the region launch: counted in [17]
    hipLaunchKernelGGL(k_partition_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, spec_of({{ST_OPT_PENDING, 1}}), h->part_lists.as<int32_t>(), (long long)n, dcnt + 16);
    HIP_TRY(h, hipGetLastError());
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], st));
    // ---- pruned masks of this level + children -------------------------------------------------------------------------------------
    if (h->mw == 2) hipLaunchKernelGGL(k_pruned_append<2>, dim3(blocks256), dim3(256), 0, st, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                       h->pruned.as<unsigned long long>() + (size_t)h->n_pruned * h->mw, ctr, keep_lowdim);
    else hipLaunchKernelGGL(k_pruned_append<4>, dim3(blocks256), dim3(256), 0, st, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                            h->pruned.as<unsigned long long>() + (size_t)h->n_pruned * h->mw, ctr, keep_lowdim);
    }
    if (gen_children) {
        HIP_TRY(h, h->childmask.ensure(nn * h->mw * sizeof(uint64_t), st));
        HIP_TRY(h, h->count.ensure(nn * sizeof(int32_t), st));
        HIP_TRY(h, h->offset.ensure(nn * sizeof(int32_t), st));
        const size_t child_bound = nn * (size_t)std::max(h->n_c - k, 1);
        HIP_TRY(h, h->children.ensure(child_bound * (k + 1) * sizeof(int32_t), st));
        HIP_TRY(h, h->parent_slot_next.ensure(child_bound * sizeof(int32_t), st));
        if (h->mw == 2) hipLaunchKernelGGL(k_children_count<2>, dim3((unsigned)n), dim3(64), 0, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                           h->pruned.as<unsigned long long>(), (long long)h->n_pruned, h->childmask.as<unsigned long long>(), h->count.as<int32_t>(), keep_lowdim);
        else hipLaunchKernelGGL(k_children_count<4>, dim3((unsigned)n), dim3(64), 0, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                h->pruned.as<unsigned long long>(), (long long)h->n_pruned, h->childmask.as<unsigned long long>(), h->count.as<int32_t>(), keep_lowdim);
        if (fused) hipLaunchKernelGGL(k_scan_publish, dim3(1), dim3(SCAN_BLOCK), 0, st, h->count.as<int32_t>(), h->offset.as<int32_t>(), (int)n, dcnt + 20,
                                      reinterpret_cast<const unsigned int *>(ctr), (int)(sizeof(LevelCounters) / 4), reinterpret_cast<const unsigned int *>(dcnt), 32, pub_dst);
        else hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SCAN_BLOCK), 0, st, h->count.as<int32_t>(), h->offset.as<int32_t>(), (int)n, dcnt + 20);
        hipLaunchKernelGGL(k_children_write, dim3((unsigned)n), dim3(64), 0, st, h->frontier.as<int32_t>(), n, k, h->mw,
                           h->childmask.as<unsigned long long>(), h->offset.as<int32_t>(), h->children.as<int32_t>(),
                           h->storing ? h->dict_stored[h->dict_cur].as<uint8_t>() : (const uint8_t *)nullptr, h->parent_slot_next.as<int32_t>());
        HIP_TRY(h, hipGetLastError());
    }
    if (!fused) hipLaunchKernelGGL(k_histogram, dim3(std::min(blocks256, 1024)), dim3(256), 0, st, h->status.as<uint8_t>(), n, ctr);
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[3], st));
    int32_t *cnt_host = h->tot_host + 16 + (int)(sizeof(LevelCounters) / 4);
    if (!fused) hipLaunchKernelGGL(k_publish_words2, dim3(1), dim3(128), 0, st, reinterpret_cast<const unsigned int *>(ctr), (int)(sizeof(LevelCounters) / 4),
                                   reinterpret_cast<const unsigned int *>(dcnt), 32, pub_dst);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(st));   // the level's only synchronisation
    LevelCounters host_ctr;
    std::memcpy(&host_ctr, h->tot_host + 16, sizeof(LevelCounters));
    h->n_smallpath++;
    small_debug_note(cnt_host);
    if (fused && (cnt_host[4] > 0 || cnt_host[24] > 0)) { h->n_smallpath_doubtful++; g_small_repeats_doubtful++; }
    if (host_ctr.n_rretry > 0 || cnt_host[17] > 0 || h->test_small_fallback || (fused && (cnt_host[4] > 0 || cnt_host[24] > 0))) {
        // a candidate k_region2 gave up on (the LDS-engine region kernel is not part of this path), or one that turned out optimal
        // after the region launch: the level is repeated classically
        h->n_smallpath_fallback++; g_small_repeats++;
        // The repeat must not look for other parents' records: this run's k_children_write has overwritten the previous level's frontier
        // (it lives in the children buffer since the hand-over), which that search walks.
        h->n_prev = 0;
        // A program that produces doubtful candidates keeps doing so (measured: 10-17 % of the small levels of random mpQPs, none on the
        // named configurations): its handle goes back to re-solving them in place, so the repeat is paid once per handle.
        if (fused && (cnt_host[4] > 0 || cnt_host[24] > 0)) h->no_small_fuse = true;
        *fallback = true;
        return MPC_OK;
    }
    float ms[3] = {0, 0, 0};
    if (h->timing) HIP_TRY(h, hipEventElapsedTime(&ms[0], h->ev[0], h->ev[1]));
    if (h->timing) HIP_TRY(h, hipEventElapsedTime(&ms[1], h->ev[1], h->ev[2]));
    if (h->timing) HIP_TRY(h, hipEventElapsedTime(&ms[2], h->ev[2], h->ev[3]));
    h->n_opt = cnt_host[6];
    h->n_children = gen_children ? cnt_host[20] : 0;
    h->n_needx = cnt_host[7];
    h->n_pruned_new = host_ctr.n_pruned_new;
    h->n_erows = host_ctr.e_rows;
    h->n_regions = (long long)host_ctr.status[ST_REGION];
    graveyard_flush(h);   // (behind the level's closing synchronisation)
    h->level_done = true;
    h->last_level_n = n;
    stream_ready(h);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n = n; stats->k = k; stats->kkt_mode = h->kkt_mode;
        for (int i = 0; i < 6; ++i) stats->n_status[i] = (int64_t)host_ctr.status[i];
        stats->n_regions = h->n_regions; stats->n_children = h->n_children; stats->n_pruned_new = h->n_pruned_new;
        stats->lp_pivots = (int64_t)host_ctr.pivots;
        stats->n_xtheta_lp = h->n_needx;
        stats->n_xtheta_fallback = (int64_t)host_ctr.xtheta_fallbacks;
        for (int i = 0; i < 4; ++i) stats->wave_cycles[i] = (int64_t)host_ctr.cycles[i];
        stats->n_x_cached = (int64_t)host_ctr.x_cached;
        stats->n_region_rows = h->n_erows;
        stats->n_opt = h->n_opt;
        stats->n_theta_items = kkc ? cnt_host[0] : n;
        if (h->n_opt > 0 && host_ctr.r2_t1 > ~host_ctr.r2_not_t0 && h->wall_khz > 0)   // k_region2 times itself on the wall clock
            stats->ms_region2 = (float)((double)(host_ctr.r2_t1 - ~host_ctr.r2_not_t0) / (double)h->wall_khz);
        stats->n_xq_items = quick_test ? cnt_host[7] : 0; stats->xq_pivots = (int64_t)host_ctr.xq_pivots;
        stats->n_x_items = (quick_test ? cnt_host[8] : cnt_host[7]) + (h->storing ? (long long)cnt_host[5] + cnt_host[6] : 0);
        stats->xq_record_ints = dict_ints_head(h->Pf.n_d0r, h->fast_x >= 2 ? 32 : 16) - 1; stats->xq_record_rows = h->Pf.n_d0r; stats->xq_record_cols = h->Pf.n_d0c + 1;
        const long long rec_bytes = (long long)(h->Pf.n_d0c + 1) * h->Pf.n_d0r * 8 + (long long)dict_ints_head(h->Pf.n_d0r, h->fast_x >= 2 ? 32 : 16) * 4;   // (what k_x2 reads of a record; the stored record also carries the children's look-up bytes)
        stats->dict_read_bytes = (h->have_prev_dict && h->have_parent_slot) ? rec_bytes : 0;
        stats->dict_write_bytes = h->storing ? rec_bytes : 0;
        stats->ms_verdict = ms[0]; stats->ms_region = ms[1]; stats->ms_children = ms[2]; stats->ms_total = ms[0] + ms[1] + ms[2];
    }
    return MPC_OK;
}

// ---- the same level for many programs at once (batch_level.hpp) -----------------------------------------------------------------
// batch_prepare = level_run_small up to its first launch (buffers, arguments), batch_finish = level_run_small after its one
// synchronisation; the launches in between are batch_level_launch's, one per stage for all members.
static int batch_prepare(mpc_handle *h, int32_t gen_children, int32_t flags, BatchMember &m) {
    const long long n = h->n;
    const int k = h->k;
    const size_t nn = (size_t)n;
    hipStream_t st = h->stream;
    std::memset(&m, 0, sizeof(m));
    h->n_opt = h->n_children = h->n_pruned_new = h->n_regions = 0;
    h->n_needx = 0;
    HIP_TRY(h, h->status.ensure(nn, st));
    HIP_TRY(h, h->pruned.ensure((size_t)(h->n_pruned + n) * h->mw * sizeof(uint64_t), st, true));
    HIP_TRY(h, h->retry_list.ensure(nn * sizeof(int32_t), st));
    HIP_TRY(h, h->theta_list.ensure(nn * sizeof(int32_t), st));
    HIP_TRY(h, h->part_lists.ensure((size_t)PART_CLASSES * nn * sizeof(int32_t), st));
    HIP_TRY(h, h->dcnt.ensure(32 * sizeof(int32_t), st));
    h->used_region2 = false; h->n_rretry = 0; h->n_erows = 0; h->rretry_rows = -1;
    h->fd = h->n_x * h->n_t + h->n_x + k * h->n_t + k;
    h->fi = 8 + k + h->n_tc + k + 2 * (h->n_c - k);
    m.k = k; m.kd = k - h->targs.ne; m.fast_t = h->fast_t; m.fast_x = h->fast_x; m.fast_r = h->fast_r; m.mw = h->mw;
    m.gen_children = gen_children ? 1 : 0;
    m.n = n; m.grid_f = h->grid_f; m.grid_r2 = h->grid_r2; m.n_cu = h->n_cu; m.lds_f = h->lds_f; m.lds_v = h->lds_v; m.lds_r2 = h->lds_r2;
    m.rsplit_max = h->rsplit_max;
    m.pf = h->pf_dev.as<DevProblem>(); m.pr = h->pr2_dev.as<DevProblem>(); m.Pv = h->Pv;
    m.fr = h->frontier.as<int32_t>(); m.status = h->status.as<uint8_t>(); m.ctr = h->ctr.as<LevelCounters>(); m.dcnt = h->dcnt.as<int32_t>();
    m.theta_list = h->theta_list.as<int32_t>(); m.retry_list = h->retry_list.as<int32_t>(); m.part_lists = h->part_lists.as<int32_t>();
    m.targs = h->targs;
    m.zero[m.n_zero++] = {h->ctr.p, sizeof(LevelCounters)};
    m.zero[m.n_zero++] = {h->dcnt.p, 32 * sizeof(int32_t)};
    m.use_kkt = (h->kkt_mode == 0 && m.kd >= 1 && m.kd <= 8 && h->no_kkt_thread != 1) ? 1 : 0;
    if (m.use_kkt && h->kkt_spread > 0 && m.kd <= BATCH_KKT_SPREAD_KMAX) m.use_kkt = 2;   // BATCH_KKT_SPREAD lanes per candidate (k_kkt_thread)
    m.kkt_listed = 0;
    if (m.use_kkt) {
        HIP_TRY(h, h->kkt_code.ensure(nn, st));
        HIP_TRY(h, h->kkt_L.ensure(nn * (size_t)k * (h->n_t + 1) * sizeof(double), st));
        m.kkt_code = h->kkt_code.as<uint8_t>(); m.kkt_L = h->kkt_L.as<double>();
        if (!h->no_kkt_lists) {   // the kernel lists what it leaves to the theta stage (as level_run_small)
            HIP_TRY(h, h->xq_list.ensure(nn * sizeof(int32_t), st));
            m.targs.kt_list = m.theta_list; m.targs.kt_n = m.dcnt + 0;
            m.targs.kx_list = h->xq_list.as<int32_t>(); m.targs.kx_n = m.dcnt + 10;
            m.kkt_listed = 1;
        }
    }
    // region stage: one slot per optimal candidate, buffers sized by the bound
    auto part_list = [&](int c) -> int32_t * { return h->part_lists.as<int32_t>() + (size_t)c * nn; };
    h->opt_ptr = part_list(2);
    {
        const int rows_t_ = h->n_c - h->n_eq + h->n_tc;
        HIP_TRY(h, h->headd.ensure(nn * h->fd * sizeof(double), st));
        HIP_TRY(h, h->headi.ensure(nn * h->fi * sizeof(int32_t), st));
        HIP_TRY(h, h->epool.ensure(nn * rows_t_ * (h->n_t + 1) * sizeof(double), st));
        const int ldk = (rows_t_ + 1 + 63) & ~63;
        HIP_TRY(h, h->kept_g.ensure(nn * ldk, st));
        HIP_TRY(h, h->done_g.ensure(nn * 2 * sizeof(unsigned int), st));
        m.zero[m.n_zero++] = {h->done_g.p, nn * 2 * sizeof(unsigned int)};
        m.rs.n_opt_dev = m.dcnt + 18;   // the region launch comes last in the batch form: all optimal candidates, class 2 of the partition at [16..19]
        m.rs.w_cap = h->grid_r2; m.rs.w_max = h->rsplit_max;
        m.W = h->no_rsplit ? 1 : 0;
        m.ldk = ldk; m.fd = h->fd; m.fi = h->fi; m.no_rbox = h->no_rbox ? 1 : 0;
        m.headd = h->headd.as<double>(); m.headi = h->headi.as<int32_t>(); m.epool = h->epool.as<double>();
        m.kept_g = h->kept_g.as<uint8_t>(); m.done_g = h->done_g.as<unsigned int>();
        h->used_region2 = true;
        // retry slots of the LDS-engine region kernel (launch_region_v1's buffers)
        m.rcap = (int)std::min<long long>(n, 256);
        HIP_TRY(h, h->recd.ensure((size_t)m.rcap * h->rec_d * sizeof(double), st));
        HIP_TRY(h, h->reci.ensure((size_t)m.rcap * h->rec_i * sizeof(int32_t), st));
        HIP_TRY(h, h->facet_flags.ensure((size_t)m.rcap * rows_t_, st));
        m.recd = h->recd.as<double>(); m.reci = h->reci.as<int32_t>(); m.facet_flags = h->facet_flags.as<uint8_t>();
        m.rec_d = h->rec_d; m.rec_i = h->rec_i; m.Pr = h->Pr; m.lds_r = h->lds_r;
    }
    // (x,theta) stage with the dictionary cache
    const int nxc = h->fast_x >= 2 ? 32 : 16;
    m.nxc = nxc;
    h->dict_stride_d = (long long)nxc * h->Pf.n_d0r;
    h->dict_stride_i = dict_ints(h->Pf.n_d0r, nxc, h->n_c);
    DictCache dc{};
    dc.fresh_limit = h->x_fresh_limit; dc.second_max = h->x_second_max;
    dc.stride_d = h->dict_stride_d; dc.stride_i = h->dict_stride_i;
    if (h->have_prev_dict && h->have_parent_slot) {
        dc.parent_slot = h->parent_slot.as<int32_t>();
        dc.prev_d = h->dict_d[1 - h->dict_cur].as<double>();
        dc.prev_i = h->dict_i[1 - h->dict_cur].as<int32_t>();
    }
    h->storing = false;
    const double need_gb = (double)nn * (h->dict_stride_d * 8.0 + h->dict_stride_i * 4.0) / 1e9;
    if (gen_children && need_gb <= h->dict_budget_gb) {
        HIP_TRY(h, h->dict_d[h->dict_cur].ensure(nn * h->dict_stride_d * sizeof(double), st));
        HIP_TRY(h, h->dict_i[h->dict_cur].ensure(nn * h->dict_stride_i * sizeof(int32_t), st));
        HIP_TRY(h, h->dict_stored[h->dict_cur].ensure(nn, st));
        m.zero[m.n_zero++] = {h->dict_stored[h->dict_cur].p, nn};
        dc.cur_d = h->dict_d[h->dict_cur].as<double>(); dc.cur_i = h->dict_i[h->dict_cur].as<int32_t>();
        dc.stored = h->dict_stored[h->dict_cur].as<uint8_t>();
        h->storing = true;
        dc.pre1 = part_list(1); dc.n_pre1_dev = m.dcnt + 5;
        dc.pre2 = part_list(2); dc.n_pre2_dev = m.dcnt + 6;
    }
    dc.chunk = 1;
    m.dc = dc;
    // one-step plans (round 5): a storing level whose candidates have parents' records; the look-up of other parents needs the previous
    // frontier, which sits in the children buffer -- only while that buffer is not about to be re-allocated for this level's children
    m.plan = 0; m.x1_buf = nullptr; m.alt = XqAlt{}; m.plan_blocks = m.x1_blocks = 0;
    if (h->storing && dc.parent_slot && h->x1 > 0 && !h->no_batch_plans && n >= h->batch_plan_min && n <= 0x7fffffffLL / 8) {
        HIP_TRY(h, h->x1_buf.ensure((6 * nn + 16) * sizeof(int32_t), st));
        m.x1_buf = h->x1_buf.as<int32_t>();
        m.plan = 1;
        const size_t child_bytes = gen_children ? nn * (size_t)std::max(h->n_c - k, 1) * (k + 1) * sizeof(int32_t) : 0;
        if (h->x1 >= 2 && h->n_prev > 0 && h->n_prev <= 0x7fffffffLL && k >= 2 && h->children.cap >= child_bytes &&
            h->children.cap >= (size_t)h->n_prev * (k - 1) * sizeof(int32_t) && h->dict_stored[1 - h->dict_cur].cap >= (size_t)h->n_prev) {
            m.alt.prev_frontier = h->children.as<int32_t>(); m.alt.prev_stored = h->dict_stored[1 - h->dict_cur].as<uint8_t>();
            m.alt.n_prev = (int)h->n_prev; m.alt.tries = MPC_MAX_NC;
        }
    }
    m.storing = h->storing ? 1 : 0;
    m.dict_stored_cur = h->storing ? h->dict_stored[h->dict_cur].as<uint8_t>() : nullptr;
    m.quick_test = (!h->storing && dc.parent_slot && !h->no_xquick) ? 1 : 0;
    // pruned masks + children
    m.keep_lowdim = (flags & MPC_LEVEL_KEEP_LOWDIM) ? 1 : 0;
    m.theta_open = h->theta_open ? 1 : 0;
    m.pruned = h->pruned.as<unsigned long long>(); m.n_pruned = h->n_pruned;
    if (gen_children) {
        HIP_TRY(h, h->childmask.ensure(nn * h->mw * sizeof(uint64_t), st));
        HIP_TRY(h, h->count.ensure(nn * sizeof(int32_t), st));
        HIP_TRY(h, h->offset.ensure(nn * sizeof(int32_t), st));
        const size_t child_bound = nn * (size_t)std::max(h->n_c - k, 1);
        HIP_TRY(h, h->children.ensure(child_bound * (k + 1) * sizeof(int32_t), st));
        HIP_TRY(h, h->parent_slot_next.ensure(child_bound * sizeof(int32_t), st));
        m.childmask = h->childmask.as<unsigned long long>(); m.count = h->count.as<int32_t>(); m.offset = h->offset.as<int32_t>();
        m.children = h->children.as<int32_t>(); m.parent_slot_next = h->parent_slot_next.as<int32_t>();
    }
    m.pub_ctr = reinterpret_cast<unsigned int *>(h->tot_dev + 16);
    m.pub_cnt = reinterpret_cast<unsigned int *>(h->tot_dev + 16 + (int)(sizeof(LevelCounters) / 4));
    // pruned.ensure(..., keep) may have moved the list: the frontier and everything else above is read after this point only
    HIP_TRY(h, hipStreamSynchronize(st));   // whatever the member's own stream still had queued (frontier kernels, copies of a grown buffer)
    return MPC_OK;
}

static int batch_finish(mpc_handle *h, int32_t gen_children, const BatchMember &m, float ms_total, mpc_level_stats *stats, bool *fallback) {
    *fallback = false;
    const long long n = h->n;
    LevelCounters host_ctr;
    std::memcpy(&host_ctr, h->tot_host + 16, sizeof(LevelCounters));
    const int32_t *cnt_host = h->tot_host + 16 + (int)(sizeof(LevelCounters) / 4);
    h->n_smallpath++;
    small_debug_note(cnt_host);
    if (cnt_host[28] > m.rcap || h->test_small_fallback) {
        // more candidates k_region2 gave up on than retry slots were reserved (never observed): the member repeats the level alone
        h->n_smallpath_fallback++;
        *fallback = true;
        return MPC_OK;
    }
    h->n_opt = cnt_host[18];
    h->n_rretry = cnt_host[28];      // their records are in the fixed layout (recd / reci), listed in retry_list
    h->n_children = gen_children ? cnt_host[20] : 0;
    h->n_needx = cnt_host[7];
    h->n_pruned_new = host_ctr.n_pruned_new;
    h->n_erows = host_ctr.e_rows;
    h->n_regions = (long long)host_ctr.status[ST_REGION];
    graveyard_flush(h);   // (behind the level's closing synchronisation)
    h->level_done = true;
    h->last_level_n = n;
    stream_ready(h);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n = n; stats->k = h->k; stats->kkt_mode = h->kkt_mode;
        for (int i = 0; i < 6; ++i) stats->n_status[i] = (int64_t)host_ctr.status[i];
        stats->n_regions = h->n_regions; stats->n_children = h->n_children; stats->n_pruned_new = h->n_pruned_new;
        stats->lp_pivots = (int64_t)host_ctr.pivots;
        stats->n_xtheta_lp = h->n_needx;
        stats->n_xtheta_fallback = (int64_t)host_ctr.xtheta_fallbacks;
        for (int i = 0; i < 4; ++i) stats->wave_cycles[i] = (int64_t)host_ctr.cycles[i];
        stats->n_x_cached = (int64_t)host_ctr.x_cached;
        stats->n_region_rows = h->n_erows;
        stats->n_opt = h->n_opt;
        stats->n_theta_items = m.use_kkt ? cnt_host[0] : n;
        stats->n_region_retry = h->n_rretry;
        if (h->n_opt > 0 && host_ctr.r2_t1 > ~host_ctr.r2_not_t0 && h->wall_khz > 0)
            stats->ms_region2 = (float)((double)(host_ctr.r2_t1 - ~host_ctr.r2_not_t0) / (double)h->wall_khz);
        stats->n_xq_items = m.quick_test ? cnt_host[7] : 0; stats->xq_pivots = (int64_t)host_ctr.xq_pivots;
        stats->n_x_items = (m.quick_test ? cnt_host[8] : cnt_host[7]) + (h->storing ? (long long)cnt_host[5] + cnt_host[6] : 0);
        stats->xq_record_ints = dict_ints_head(h->Pf.n_d0r, h->fast_x >= 2 ? 32 : 16) - 1; stats->xq_record_rows = h->Pf.n_d0r; stats->xq_record_cols = h->Pf.n_d0c + 1;
        const long long rec_bytes = (long long)(h->Pf.n_d0c + 1) * h->Pf.n_d0r * 8 + (long long)dict_ints_head(h->Pf.n_d0r, h->fast_x >= 2 ? 32 : 16) * 4;   // (what k_x2 reads of a record; the stored record also carries the children's look-up bytes)
        stats->dict_read_bytes = (h->have_prev_dict && h->have_parent_slot) ? rec_bytes : 0;
        stats->dict_write_bytes = h->storing ? rec_bytes : 0;
        stats->ms_total = ms_total;   // the batch's launches, first to last (shared by all members)
    }
    return MPC_OK;
}

static int level_run_impl(mpc_handle *h, int32_t gen_children, int32_t flags, mpc_level_stats *stats);

// device memory one member's level holds on the no-round-trip path (the buffers are sized by the bound n): region records, children,
// two generations of the dictionary cache
static double batch_level_gb(const mpc_handle *h, int32_t gen_children) {
    const double n = (double)h->n;
    const int k = h->k;
    const double rows_t = h->n_c - h->n_eq + h->n_tc;
    double bytes = n * ((double)(h->n_x * h->n_t + h->n_x + k * h->n_t + k) * 8.0 + (double)(8 + 2 * k + h->n_tc + 2 * (h->n_c - k)) * 4.0 + rows_t * (h->n_t + 1) * 8.0);
    if (gen_children) bytes += n * std::max(h->n_c - k, 1) * (k + 2) * 4.0;
    const double nxc = h->fast_x >= 2 ? 32 : 16;
    bytes += 2.0 * n * (nxc * h->Pf.n_d0r * 8.0 + (2.0 * h->Pf.n_d0r + nxc + 4) * 4.0);
    return bytes / 1e9;
}

struct BatchToken {
    std::vector<mpc_handle *> hs;
    std::vector<int32_t> gen;
    int32_t flags = 0;
    std::vector<BatchMember> members;
    std::vector<int> alone;      // members outside the shared launches: run one after the other by mpc_level_run's own paths
    hipStream_t st = nullptr;
    DevBuf tab_dev;              // the members' argument table the launches read (one per batch call: concurrent batches do not share it)
    HostBuf tab_host;
    ~BatchToken() { tab_dev.release(); tab_host.release(); }   // only after the launches have completed (wait) or were never queued
};

int mpc_level_batch_start(mpc_handle **hs, int32_t n_handles, const int32_t *gen_children, int32_t flags, void **token) {
    if (!hs || n_handles <= 0 || !gen_children || !token) return MPC_ERR_INVALID;
    *token = nullptr;
    mpc_handle *h0 = hs[0];
    if (!h0) return MPC_ERR_INVALID;
    for (int i = 0; i < n_handles; ++i) {
        mpc_handle *h = hs[i];
        if (!h) return MPC_ERR_INVALID;
        if (h->device != h0->device) return fail(h, MPC_ERR_INVALID, "mpc_level_run_batch: the members must live on one device");
        if (flags & MPC_LEVEL_GRAPH) return fail(h, MPC_ERR_INVALID, "mpc_level_run_batch: MPC_LEVEL_GRAPH is not a batch level");
        { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "a level started with mpc_level_start is still running"); }
        { std::lock_guard<std::mutex> lk(h->wm); h->w_stream_ready = false; h->a_ready.store(0, std::memory_order_release); }
    }
    {
        std::vector<mpc_handle *> sorted(hs, hs + n_handles);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(h0, MPC_ERR_INVALID, "mpc_level_run_batch: a handle appears twice");
    }
    HIP_TRY(h0, hipSetDevice(h0->device));
    if (fetch_many_wait(h0->device) != MPC_OK) return fail(h0, MPC_ERR_HIP, "mpc_level_batch_start: the shared record copy (k_fetch_many) failed");   // the members' record buffers are free again
    std::unique_ptr<BatchToken> t(new BatchToken());
    t->hs.assign(hs, hs + n_handles);
    t->gen.assign(gen_children, gen_children + n_handles);
    t->flags = flags & ~(MPC_LEVEL_THEN_BASE | MPC_LEVEL_ONLY_BASE | MPC_LEVEL_STREAM);
    t->members.reserve(n_handles);
    const double budget_gb = [] { const char *ev = std::getenv("MPC_BATCH_BUDGET_GB"); return ev ? std::atof(ev) : 160.0; }();
    double used_gb = 0.0;
    for (int i = 0; i < n_handles; ++i) {
        mpc_handle *h = hs[i];
        stream_release(h);
        { int rcj = x1_join(h); if (rcj) return rcj; }
        // MPC_SMALLPATH_MAX bounds the single-program path only (above it the overlapped classic path is faster for ONE program)
        const long long keep_max = h->smallpath_max;
        // (bounded: the shared launches compact / partition / scan a member with single 1024-thread blocks -- beyond 2^18 candidates a
        // member takes its turn alone, on the multi-block kernels of the classic path)
        h->smallpath_max = std::max<long long>(keep_max, 1LL << 18);
        const bool ok = small_path_ok(h, h->n, h->k, t->flags, gen_children[i]);
        h->smallpath_max = keep_max;
        bool ok_i = ok;
        if (ok_i) {
            // MPC_BATCH_BUDGET_GB (default 160 of the 288 GB): members beyond it take their turn after the shared launches, one at a time
            const double gb = batch_level_gb(h, gen_children[i]);
            if (used_gb + gb > budget_gb && !t->members.empty()) ok_i = false; else used_gb += gb;
        }
        if (!ok_i) { t->alone.push_back(i); HIP_TRY(h, hipStreamSynchronize(h->stream)); continue; }   // (copies of the last level's records: complete when this call returns, as for the members)
        BatchMember m;
        const int rc = batch_prepare(h, gen_children[i], t->flags, m);
        if (rc != MPC_OK) return rc;
        m.id = i;
        t->members.push_back(m);
    }
    if (!t->members.empty()) {
        mpc_handle *hl = hs[t->members[0].id];     // ids are the caller's positions (batch_level_launch reorders the members)
        t->st = hl->stream;
        const int lead = t->members[0].id;
        if (hl->timing) HIP_TRY(hl, hipEventRecord(hl->ev[0], t->st));
        const size_t tab_bytes = t->members.size() * sizeof(BatchMember);
        HIP_TRY(hl, t->tab_dev.ensure(tab_bytes, t->st));
        HIP_TRY(hl, t->tab_host.ensure(tab_bytes));
        hipError_t e = batch_level_launch(t->members.data(), (int)t->members.size(), t->st, t->tab_host.as<BatchMember>(), t->tab_dev.as<BatchMember>());
        if (e != hipSuccess) { (void)hipStreamSynchronize(t->st); return fail(hl, MPC_ERR_HIP, std::string("batch level launch: ") + hipGetErrorString(e)); }   // (what was queued still reads the table)
        if (hl->timing) HIP_TRY(hl, hipEventRecord(hl->ev[3], t->st));
        t->alone.insert(t->alone.begin(), -1 - lead);      // first entry < 0: the member whose events bracket the launches
    }
    *token = t.release();
    return MPC_OK;
}

int mpc_level_batch_wait(void *token, mpc_level_stats *stats, int32_t *n_batched) {
    if (!token) return MPC_ERR_INVALID;
    std::unique_ptr<BatchToken> t(static_cast<BatchToken *>(token));
    if (n_batched) *n_batched = 0;
    if (!t->members.empty()) {
        const int lead = -1 - t->alone.front();
        t->alone.erase(t->alone.begin());
        mpc_handle *hl = t->hs[lead];
        HIP_TRY(hl, hipSetDevice(hl->device));
        HIP_TRY(hl, hipStreamSynchronize(t->st));   // the level's only synchronisation, for all members
        float ms = 0;
        if (hl->timing) HIP_TRY(hl, hipEventElapsedTime(&ms, hl->ev[0], hl->ev[3]));
        for (const BatchMember &m : t->members) {
            bool fallback = false;
            const int rc = batch_finish(t->hs[m.id], t->gen[m.id], m, ms, stats ? stats + m.id : nullptr, &fallback);
            if (rc != MPC_OK) return rc;
            if (fallback) {
                // The member ran in the shared launches: m_children_write has overwritten the previous level's frontier in h->children,
                // which the classic path's search for other parents (k_xq_thread, alt.prev_frontier) would walk with the wrong row width.
                // As in level_run_small's repeat: no other-parent look-ups on the repeated level.  (Members that never ran keep n_prev.)
                t->hs[m.id]->n_prev = 0;
                t->alone.push_back(m.id);
            }
            else if (n_batched) ++*n_batched;
        }
    }
    for (int i : t->alone) {
        mpc_handle *h = t->hs[i];
        h->skip_small = true;
        const int rc = level_run_impl(h, t->gen[i], t->flags, stats ? stats + i : nullptr);
        h->skip_small = false;
        if (rc != MPC_OK) return rc;
    }
    return MPC_OK;
}

int mpc_level_run_batch(mpc_handle **hs, int32_t n_handles, const int32_t *gen_children, int32_t flags, mpc_level_stats *stats, int32_t *n_batched) {
    void *token = nullptr;
    const int rc = mpc_level_batch_start(hs, n_handles, gen_children, flags, &token);
    if (rc != MPC_OK) return rc;
    return mpc_level_batch_wait(token, stats, n_batched);
}

static int level_run_impl(mpc_handle *h, int32_t gen_children, int32_t flags, mpc_level_stats *stats) {
    if (!h) return MPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    stream_release(h);
    // (a deferred k_x1 of the previous level: joined at once unless this is a large level whose first kernel, k_kkt_thread, reads no dictionary)
    const bool x1_late_join = h->x1_pending && h->fast && !h->force_v1 && h->kkt_mode == 0 && h->no_kkt_thread != 1 && !h->no_lean &&
                              (h->skip_small || !small_path_ok(h, h->n, h->k, flags, gen_children)) && h->k - h->targs.ne >= 1 && h->k - h->targs.ne <= KKT_THREAD_MAX;
    if (!x1_late_join) { int rcj = x1_join(h); if (rcj) return rcj; }
    if (!h->skip_small && small_path_ok(h, h->n, h->k, flags, gen_children)) {
        bool fallback = false;
        const int rcs = level_run_small(h, gen_children, flags, stats, &fallback);
        if (rcs != MPC_OK || !fallback) return rcs;
    }
    const long long n = h->n;
    const int k = h->k;
    hipStream_t st = h->stream;
    h->region_side_stream = false;
    h->n_opt = h->n_children = h->n_pruned_new = h->n_regions = 0;
    h->n_needx = 0;
    LevelCounters host_ctr;
    std::memset(&host_ctr, 0, sizeof(host_ctr));
    float ms[3] = {0, 0, 0}, kms[5] = {0, 0, 0, 0, 0};
    bool kernel_timed[5] = {false, false, false, false, false};
    long long n_x_items = 0, n_theta_items = 0, n_xq_items = 0;
    bool xq_thread_timed = false, xq_early_ran = false, x1_ran = false;
    h->n_xq_thread = 0; h->ms_xq_thread = 0;
    if (n > 0) {
        const size_t nn = (size_t)n;
        HIP_TRY(h, h->status.ensure(nn, st));
        HIP_TRY(h, h->flag.ensure(nn * sizeof(int32_t), st));
        HIP_TRY(h, h->pos.ensure(nn * sizeof(int32_t), st));
        HIP_TRY(h, h->opt_list.ensure(nn * sizeof(int32_t), st));
        HIP_TRY(h, h->pruned.ensure((size_t)(h->n_pruned + n) * h->mw * sizeof(uint64_t), st, true));   // the level's newly pruned masks go behind the list
        LevelCounters *ctr = h->ctr.as<LevelCounters>();
        int32_t *total = h->tot_dev;          // device alias of h->tot_host: valid on the host after the next synchronisation
        const bool small = n <= SMALL_LEVEL_N;
        const int blocks256 = (int)((n + 255) / 256);
        // compacts the candidates whose status lies in [lo, hi] into h->retry_list; returns their number
        // count_dev != nullptr: the length goes to device memory only and nobody waits for it (the consumer kernel reads it there)
        auto compact = [&](int lo, int hi, int32_t *count, int32_t *count_dev = nullptr) -> int {
            HIP_TRY(h, h->retry_list.ensure(nn * sizeof(int32_t), st));
            int32_t *tot = count_dev ? count_dev : total;
            if (small) {
                hipLaunchKernelGGL(k_compact_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, lo, hi, h->retry_list.as<int32_t>(), tot);
            } else {
                hipLaunchKernelGGL(k_flag_status, dim3(blocks256), dim3(256), 0, st, h->status.as<uint8_t>(), n, lo, hi, h->flag.as<int32_t>());
                int rcs = launch_scan(h, h->flag.as<int32_t>(), h->pos.as<int32_t>(), n, tot);
                if (rcs) return rcs;
                hipLaunchKernelGGL(k_scatter_index, dim3(blocks256), dim3(256), 0, st, h->flag.as<int32_t>(), h->pos.as<int32_t>(), n, h->retry_list.as<int32_t>());
            }
            HIP_TRY(h, hipGetLastError());
            if (count_dev) return MPC_OK;
            HIP_TRY(h, hipStreamSynchronize(st));
            *count = h->tot_host[0];
            return MPC_OK;
        };
        // Lean form of a large level (round 3): the classic sequence read a list length back after almost every stage -- 22 idle gaps of
        // 15-30 us per solve of config 4 (tools/gpu_idle.sh).  The host needs only the lengths that size the region stage (one
        // read-back after the theta stage) and the final statistics; the theta LP, the (x,theta) stage after the quick test and the
        // child generation take theirs from device memory (dcnt: [0] theta list, [8] open after the quick test, [20] children) and are
        // launched for the bound the host does know.
        const bool lean = !h->no_lean;
        HIP_TRY(h, h->dcnt.ensure(32 * sizeof(int32_t), st));
        {   // counters and list lengths cleared by one launch (a hipMemsetAsync costs the host ~16 us, a launch ~3 us)
            ZeroBufs z{};
            z.p[0] = h->ctr.p; z.bytes[0] = sizeof(LevelCounters);
            z.p[1] = h->dcnt.p; z.bytes[1] = 32 * sizeof(int32_t);
            hipLaunchKernelGGL(k_zero_bufs, dim3(1), dim3(256), 0, st, z);
            HIP_TRY(h, hipGetLastError());
        }
        // (round 6) the pruned list of the earlier levels, bucketed by smallest non-equality member, for this level's children stage: three
        // small launches at the level's start (nothing waits for them before k_children_count_b)
        bool pruned_bucketed = false;
        if (gen_children && !(flags & MPC_LEVEL_GRAPH) && h->pruned_bucket_min > 0 && h->n_c <= 256 && h->n_pruned >= std::max<long long>(1, h->pruned_bucket_np) &&
            (double)n * (double)h->n_pruned >= h->pruned_bucket_min && h->n_pruned <= 0x7fffffffLL) {
            HIP_TRY(h, h->pruned_b.ensure((size_t)h->n_pruned * h->mw * sizeof(uint64_t), st));
            HIP_TRY(h, h->pruned_head.ensure((size_t)PB_WORDS * sizeof(int32_t), st));
            HIP_TRY(h, hipMemsetAsync(h->pruned_head.p, 0, (size_t)PB_WORDS * sizeof(int32_t), st));
            const dim3 gb((unsigned)((h->n_pruned + 255) / 256)), bb(256);
            const int ne_b = h->n_eq;
            if (h->mw == 2) hipLaunchKernelGGL(k_pruned_bucket_count<2>, gb, bb, 0, st, h->pruned.as<unsigned long long>(), (long long)h->n_pruned, ne_b, h->pruned_head.as<int32_t>());
            else hipLaunchKernelGGL(k_pruned_bucket_count<4>, gb, bb, 0, st, h->pruned.as<unsigned long long>(), (long long)h->n_pruned, ne_b, h->pruned_head.as<int32_t>());
            hipLaunchKernelGGL(k_pruned_bucket_scan, dim3(1), dim3(256), 0, st, h->pruned_head.as<int32_t>());
            if (h->mw == 2) hipLaunchKernelGGL(k_pruned_bucket_scatter<2>, gb, bb, 0, st, h->pruned.as<unsigned long long>(), (long long)h->n_pruned, ne_b, h->pruned_head.as<int32_t>(), h->pruned_b.as<unsigned long long>());
            else hipLaunchKernelGGL(k_pruned_bucket_scatter<4>, gb, bb, 0, st, h->pruned.as<unsigned long long>(), (long long)h->n_pruned, ne_b, h->pruned_head.as<int32_t>(), h->pruned_b.as<unsigned long long>());
            HIP_TRY(h, hipGetLastError());
            pruned_bucketed = true;
        }
        // (round 6) a large last level: the theta kernel will list its optimal candidates itself (the queue form below); entries -1 = not written
        const bool r2_queue_ready = lean && h->r2_early > 0 && !gen_children && n >= h->r2_early_min && n <= 0x7fffffffLL && h->fast && h->fast_r >= 0 && !h->force_v1 && !(flags & MPC_LEVEL_GRAPH);
        if (r2_queue_ready) {
            hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)std::min<long long>(1024, (n + 255) / 256)), dim3(256), 0, st, h->opt_list.as<int32_t>(), n, -1);
            HIP_TRY(h, hipGetLastError());
        }
        int32_t *dcnt = h->dcnt.as<int32_t>();
        bool theta_lean = false, xq_lean = false, children_lean = false;
        // deterministic partition of the candidates into up to four lists by status (spec: status -> class nibble, 15 = none);
        // the lists are h->part_lists + c * n, their lengths come back in counts[]
        const int nb1024 = (int)((n + 1023) / 1024);
        // deferred: the counts go to the pinned words [12..15] and nobody waits here (the caller reads them after a later synchronisation)
        auto partition = [&](std::initializer_list<std::pair<int, int>> classes, int32_t counts[PART_CLASSES], bool deferred = false, int32_t *tot_dev_mem = nullptr) -> int {
            unsigned long long spec = ~0ull;
            for (const auto &sc : classes) spec = (spec & ~(15ull << (4 * sc.first))) | ((unsigned long long)sc.second << (4 * sc.first));
            HIP_TRY(h, h->part_counts.ensure((size_t)PART_CLASSES * nb1024 * sizeof(int32_t), st));
            HIP_TRY(h, h->part_lists.ensure((size_t)PART_CLASSES * nn * sizeof(int32_t), st));
            int32_t *tot = tot_dev_mem ? tot_dev_mem : (deferred ? total + 12 : total);   // tot_dev_mem: the lengths stay in device memory (the caller publishes them)
            if (small) {
                hipLaunchKernelGGL(k_partition_small, dim3(1), dim3(1024), 0, st, h->status.as<uint8_t>(), (int)n, spec, h->part_lists.as<int32_t>(), (long long)n, tot);
            } else {
                if (h->helper_it == 4) {
                    hipLaunchKernelGGL(k_part_count<4>, dim3(nb1024), dim3(256), 0, st, h->status.as<uint8_t>(), n, spec, h->part_counts.as<int32_t>(), nb1024);
                    hipLaunchKernelGGL(k_part_sums<4>, dim3(PART_CLASSES), dim3(256), 0, st, h->part_counts.as<int32_t>(), nb1024, tot);
                    hipLaunchKernelGGL(k_part_scatter<4>, dim3(nb1024), dim3(256), 0, st, h->status.as<uint8_t>(), n, spec, h->part_counts.as<int32_t>(), nb1024,
                                       h->part_lists.as<int32_t>());
                } else {
                    hipLaunchKernelGGL(k_part_count<1>, dim3(nb1024), dim3(1024), 0, st, h->status.as<uint8_t>(), n, spec, h->part_counts.as<int32_t>(), nb1024);
                    hipLaunchKernelGGL(k_part_sums<1>, dim3(PART_CLASSES), dim3(1024), 0, st, h->part_counts.as<int32_t>(), nb1024, tot);
                    hipLaunchKernelGGL(k_part_scatter<1>, dim3(nb1024), dim3(1024), 0, st, h->status.as<uint8_t>(), n, spec, h->part_counts.as<int32_t>(), nb1024,
                                       h->part_lists.as<int32_t>());
                }
            }
            HIP_TRY(h, hipGetLastError());
            if (deferred) return MPC_OK;
            HIP_TRY(h, hipStreamSynchronize(st));
            for (int c = 0; c < PART_CLASSES; ++c) counts[c] = h->tot_host[c];
            return MPC_OK;
        };
        // clears / copies / slot marks that a region launch and the (x,theta) stage need, collected and issued as ONE launch each
        // (k_region_prep): `prep` for the main stream, `rprep` for what only the region kernel reads (issued on ITS stream)
        RegionPrep prep{}, rprep{};
        bool prep_any = false, rprep_any = false;
        auto prep_flush_on = [&](RegionPrep &pp, bool &any, hipStream_t ps) -> int {
            if (!any) return MPC_OK;
            unsigned long long work = (unsigned long long)std::max<long long>(pp.copy_n, pp.extra);
            for (int j = 0; j < 4; ++j) work = std::max(work, pp.z.bytes[j] / 8);
            hipLaunchKernelGGL(k_region_prep, dim3((unsigned)std::min<unsigned long long>(256, work / 1024 + 1)), dim3(256), 0, ps, pp);
            HIP_TRY(h, hipGetLastError());
            pp = RegionPrep{}; any = false;
            return MPC_OK;
        };
        auto prep_flush = [&]() -> int { return prep_flush_on(prep, prep_any, st); };
        auto prep_zero_in = [&](RegionPrep &pp, bool &any, hipStream_t ps, void *ptr, size_t bytes) -> int {
            if (!ptr || bytes == 0) return MPC_OK;
            int j = 0;
            while (j < 4 && pp.z.p[j]) ++j;
            if (j == 4) { int rcs = prep_flush_on(pp, any, ps); if (rcs) return rcs; j = 0; }
            pp.z.p[j] = ptr; pp.z.bytes[j] = bytes; any = true;
            return MPC_OK;
        };
        auto prep_zero = [&](void *ptr, size_t bytes) -> int { return prep_zero_in(prep, prep_any, st, ptr, bytes); };
        auto part_list = [&](int c) -> int32_t * { return h->part_lists.as<int32_t>() + (size_t)c * nn; };
        const uint8_t *kkc = nullptr;   // KKT codes and multipliers of k_kkt_thread (read by k_theta2 / k_region2)
        const double *kkl = nullptr;
        // region stage on the register engine: one slot per candidate of h->opt_ptr.  Buffers are prepared on the main stream; the
        // kernel goes to `rst` (the main stream, or stream3 when the stage runs under the level's (x,theta) stage)
        h->used_region2 = false; h->n_rretry = 0; h->n_erows = 0; h->rretry_rows = -1;
        h->fd = h->n_x * h->n_t + h->n_x + k * h->n_t + k;
        h->fi = 8 + k + h->n_tc + k + 2 * (h->n_c - k);
        int32_t *region_out_hi = nullptr;   // head_i of the level's slots as the device sees it (device buffer or mapped host block)
        // Round 6, the queue form (`early`): the launch runs BESIDE the theta kernel and takes the optimal candidates from the queue that kernel
        // fills (n_opt is then the BOUND that sizes the buffers: the length of the theta list); region2_drain below is its second launch.
        struct R2Saved { double *hd = nullptr, *er = nullptr; int32_t *hi = nullptr; RegionStream rs{}; int ldk = 0; bool valid = false; } r2s;
        auto region2_launch = [&](int32_t n_opt, int32_t extra, hipStream_t rst, bool one_wave, bool early = false) -> int {
            const int rows_t_ = h->n_c - h->n_eq + h->n_tc;
            const size_t n_tot = (size_t)n_opt + (size_t)extra;   // slots: the launch's candidates + spare ones for late optimal candidates
            HIP_TRY(h, h->headd.ensure(n_tot * h->fd * sizeof(double), st));
            HIP_TRY(h, h->headi.ensure(n_tot * h->fi * sizeof(int32_t), st));
            HIP_TRY(h, h->epool.ensure(n_tot * rows_t_ * (h->n_t + 1) * sizeof(double), st));
            // few optimal candidates: several wavefronts per candidate (the facet tests are split among them)
            int W = (h->no_rsplit || one_wave || early) ? 1 : h->rsplit_max;
            while (W > 1 && (long long)n_opt * W > h->grid_r2) W >>= 1;
            const int ldk = (rows_t_ + 1 + 63) & ~63;
            if (W > 1) {
                HIP_TRY(h, h->kept_g.ensure((size_t)n_opt * ldk, st));
                HIP_TRY(h, h->done_g.ensure((size_t)n_opt * 2 * sizeof(unsigned int), st));
                { int rcs = prep_zero_in(rprep, rprep_any, rst, h->done_g.p, (size_t)n_opt * 2 * sizeof(unsigned int)); if (rcs) return rcs; }
            }
            // an overlapped launch may take only a share of the wave slots (r2_cap_pct): its 256-register wavefronts otherwise fill the
            // register file of every SIMD they sit on and the (x,theta) kernel beside them gets no slot there until they leave
            const long long grid_cap = early ? std::min<long long>(h->grid_r2, (long long)h->n_cu * h->r2_early_wpc)
                                       : ((rst != st && one_wave) ? std::max<long long>(h->n_cu, (long long)h->grid_r2 * h->r2_cap_pct / 100) : h->grid_r2);
            const dim3 g((unsigned)std::min<long long>((long long)n_opt * W, grid_cap)), b(64);
            const DevProblem *pr = h->pr2_dev.as<DevProblem>();
            // where the records go: device buffers (fetched / gathered later), or -- streaming -- page-locked host blocks the
            // kernel writes directly, in chunks the host consumes while the kernel is still running
            double *out_hd = h->headd.as<double>(), *out_er = h->epool.as<double>();
            int32_t *out_hi = h->headi.as<int32_t>();
            RegionStream rs{};
            const size_t bytes_hd = n_tot * h->fd * sizeof(double), bytes_hi = n_tot * h->fi * sizeof(int32_t),
                         bytes_er = n_tot * rows_t_ * (h->n_t + 1) * sizeof(double);
            if ((flags & MPC_LEVEL_STREAM) && bytes_hd + bytes_hi + bytes_er <= (size_t(1) << 30)) {
                auto &so = h->so;
                so.shift = 8;
                while (so.shift > 4 && ((long long)n_opt >> so.shift) < 8) --so.shift;   // at least ~8 chunks, 16..256 slots each
                so.n_chunks = (int)(((long long)n_opt + (1ll << so.shift) - 1) >> so.shift);
                so.n_slots = (long long)n_tot; so.cap_rows = (long long)n_tot * rows_t_;   // chunks cover the first n_opt slots
                HIP_TRY(h, host_pool_take(bytes_hd, &so.hd, nullptr, true));
                HIP_TRY(h, host_pool_take(bytes_hi, &so.hi, nullptr, true));
                HIP_TRY(h, host_pool_take(std::max<size_t>(bytes_er, 8), &so.er, nullptr, true));
                h->st_flags.coherent = true;
                HIP_TRY(h, h->st_flags.ensure((size_t)so.n_chunks * sizeof(int32_t)));
                std::memset(h->st_flags.p, 0, (size_t)so.n_chunks * sizeof(int32_t));
                h->cw_chunks = so.n_chunks;
                HIP_TRY(h, h->chunk_count.ensure((size_t)so.n_chunks * sizeof(unsigned int), st));
                { int rcs = prep_zero_in(rprep, rprep_any, rst, h->chunk_count.p, (size_t)so.n_chunks * sizeof(unsigned int)); if (rcs) return rcs; }
                void *d_hd = nullptr, *d_hi = nullptr, *d_er = nullptr, *d_fl = nullptr;
                HIP_TRY(h, hipHostGetDevicePointer(&d_hd, so.hd, 0));
                HIP_TRY(h, hipHostGetDevicePointer(&d_hi, so.hi, 0));
                HIP_TRY(h, hipHostGetDevicePointer(&d_er, so.er, 0));
                HIP_TRY(h, hipHostGetDevicePointer(&d_fl, h->st_flags.p, 0));
                out_hd = static_cast<double *>(d_hd); out_hi = static_cast<int32_t *>(d_hi); out_er = static_cast<double *>(d_er);
                rs.count = h->chunk_count.as<unsigned int>(); rs.flags = static_cast<int32_t *>(d_fl); rs.shift = so.shift; rs.n_slots = n_opt;
                so.active = true;
            }
#define MPC_LAUNCH_R2(NT_, SL_) hipLaunchKernelGGL((k_region2<NT_, SL_>), g, b, h->lds_r2, rst, pr, h->frontier.as<int32_t>(), k, h->opt_ptr, n_opt, \
                                                   h->status.as<uint8_t>(), out_hd, out_hi, h->fd, h->fi, out_er, ctr, kkc, kkl, \
                                                   W, h->kept_g.as<uint8_t>(), ldk, h->done_g.as<unsigned int>(), \
                                                   h->no_rbox ? (const double *)nullptr : h->targs.tvp + (size_t)NT_ * NT_ + NT_, rs)
            region_out_hi = out_hi;
            if (early) {
                // every slot is marked empty first: which of them the queue will reach is not known yet (the host's region objects are cut
                // from whole chunks of slots)
                rprep.head_i = out_hi; rprep.fi = h->fi; rprep.first = 0; rprep.extra = (int)n_tot; rprep_any = true;
                rs.early = 1; rs.spin_max = h->r2_early_spin; rs.q_cap = (int)std::min<size_t>(nn, 0x7fffffff);
                r2s.hd = out_hd; r2s.hi = out_hi; r2s.er = out_er; r2s.rs = rs; r2s.ldk = ldk; r2s.valid = true;
            } else if (extra > 0) { rprep.head_i = out_hi; rprep.fi = h->fi; rprep.first = n_opt; rprep.extra = extra; rprep_any = true; }
            if (early) {
                // (everything this launch reads was queued on the main stream in front of the event the host has just waited for)
                { int rcs = prep_flush_on(rprep, rprep_any, rst); if (rcs) return rcs; }
            } else if (rst != st) {
                // The side stream does NOT wait for the main stream: every caller has synchronised the main stream (the partition whose
                // counts sized this launch) and has queued nothing since that the region kernel reads, so the region stage's own
                // preparation and the kernel go straight to the side stream -- one cross-stream hop (~25 us) less per large level than the
                // fork through an event.  The main stream continues only when the side stream has reached the region kernel, so that
                // the region wavefronts (the long chains) are placed first and the (x,theta) kernels fill in around them -- without this
                // the two dispatches race and the persistent (x,theta) kernel often takes the whole GPU first.
                if (h->r3_fork_event) {   // MPC_R3_FORK=1: round-3 form, the side stream starts behind an event of the main stream
                    HIP_TRY(h, hipEventRecord(h->ev_rfork, st));
                    HIP_TRY(h, hipStreamWaitEvent(rst, h->ev_rfork, 0));
                }
                { int rcs = prep_flush_on(rprep, rprep_any, rst); if (rcs) return rcs; }
                HIP_TRY(h, hipEventRecord(h->ev_rgo, rst));
                HIP_TRY(h, hipStreamWaitEvent(st, h->ev_rgo, 0));
            } else {
                { int rcs = prep_flush(); if (rcs) return rcs; }
                { int rcs = prep_flush_on(rprep, rprep_any, st); if (rcs) return rcs; }
            }
            if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[4], rst));
            if (early && h->r2_early_wpc <= 0) { /* MPC_R2_EARLY_WPC=0: no wavefronts beside the theta kernel, the drain launch alone (it still starts without a partition and its read-back) */ }
            else
            switch (h->fast_r) {
                case 0: MPC_LAUNCH_R2(4, 1); break;
                case 1: MPC_LAUNCH_R2(4, 2); break;
                case 2: MPC_LAUNCH_R2(8, 1); break;
                case 3: MPC_LAUNCH_R2(8, 2); break;
                case 4: MPC_LAUNCH_R2(10, 1); break;
                default: MPC_LAUNCH_R2(10, 2); break;
            }
#undef MPC_LAUNCH_R2
            if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[5], rst));
            if (rst != st) { h->r3_dirty = true; h->region_side_stream = true; HIP_TRY(h, hipEventRecord(h->ev_rjoin, rst)); }
            kernel_timed[2] = true;
            HIP_TRY(h, hipGetLastError());
            h->used_region2 = true;
            if (!early) stream_ready(h);   // the caller of mpc_level_stream_info may start consuming chunks (queue form: once the queue's length is known)
            return MPC_OK;
        };
        // the drain launch of the queue form: behind the theta kernel (the queue is closed), on its own stream beside the early launch
        auto region2_drain = [&](int32_t n_opt) -> int {
            const dim3 g((unsigned)std::max<long long>(1, std::min<long long>(n_opt, h->grid_r2))), b(64);
            const DevProblem *pr = h->pr2_dev.as<DevProblem>();
            RegionStream rs = r2s.rs;
            rs.early = 2;
            double *out_hd = r2s.hd, *out_er = r2s.er; int32_t *out_hi = r2s.hi;
            const int ldk = r2s.ldk, W = 1;
            hipStream_t rst = h->stream4;
#define MPC_LAUNCH_R2(NT_, SL_) hipLaunchKernelGGL((k_region2<NT_, SL_>), g, b, h->lds_r2, rst, pr, h->frontier.as<int32_t>(), k, h->opt_ptr, n_opt, \
                                                   h->status.as<uint8_t>(), out_hd, out_hi, h->fd, h->fi, out_er, ctr, kkc, kkl, \
                                                   W, h->kept_g.as<uint8_t>(), ldk, h->done_g.as<unsigned int>(), \
                                                   h->no_rbox ? (const double *)nullptr : h->targs.tvp + (size_t)NT_ * NT_ + NT_, rs)
            switch (h->fast_r) {
                case 0: MPC_LAUNCH_R2(4, 1); break;
                case 1: MPC_LAUNCH_R2(4, 2); break;
                case 2: MPC_LAUNCH_R2(8, 1); break;
                case 3: MPC_LAUNCH_R2(8, 2); break;
                case 4: MPC_LAUNCH_R2(10, 1); break;
                default: MPC_LAUNCH_R2(10, 2); break;
            }
#undef MPC_LAUNCH_R2
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipEventRecord(h->ev_rjoin2, rst));
            return MPC_OK;
        };
        bool r2_early = false, r2_drained = false, r2_early_A = false;   // r2_early: the theta kernel lists its optimal candidates; _A: an early region launch runs beside it
        long long r2_early_ntot = 0;   // slots of the queue form's buffers (the bound + spare ones)
        bool region_launched = false;
        int32_t n_late = 0;   // optimal candidates found after an overlapped region launch (spare slots)
        int32_t n_opt_fast = -1;   // >= 0: the fast path has already built h->opt_list
        // The end of a level -- pruned masks, children, histogram, counters published -- as launches only (the caller synchronises).
        bool tail_done = false;
        auto queue_tail = [&]() -> int {
            const int keep_lowdim = (flags & MPC_LEVEL_KEEP_LOWDIM) ? 1 : 0;
            if (flags & MPC_LEVEL_GRAPH) { /* no pruning in the graph traversal */ }
            else if (h->mw == 2) hipLaunchKernelGGL(k_pruned_append<2>, dim3(blocks256), dim3(256), 0, st, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                               h->pruned.as<unsigned long long>() + (size_t)h->n_pruned * h->mw, ctr, keep_lowdim);
            else hipLaunchKernelGGL(k_pruned_append<4>, dim3(blocks256), dim3(256), 0, st, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                    h->pruned.as<unsigned long long>() + (size_t)h->n_pruned * h->mw, ctr, keep_lowdim);
            if (gen_children) {
                HIP_TRY(h, h->childmask.ensure(nn * h->mw * sizeof(uint64_t), st));
                HIP_TRY(h, h->count.ensure(nn * sizeof(int32_t), st));
                HIP_TRY(h, h->offset.ensure(nn * sizeof(int32_t), st));
                if (pruned_bucketed) {
                    if (h->mw == 2) hipLaunchKernelGGL(k_children_count_b<2>, dim3((unsigned)n), dim3(64), 0, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                                       h->pruned_b.as<unsigned long long>(), h->pruned_head.as<int32_t>(), h->childmask.as<unsigned long long>(), h->count.as<int32_t>(), keep_lowdim);
                    else hipLaunchKernelGGL(k_children_count_b<4>, dim3((unsigned)n), dim3(64), 0, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                            h->pruned_b.as<unsigned long long>(), h->pruned_head.as<int32_t>(), h->childmask.as<unsigned long long>(), h->count.as<int32_t>(), keep_lowdim);
                } else
                if (h->mw == 2) hipLaunchKernelGGL(k_children_count<2>, dim3((unsigned)n), dim3(64), 0, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                                   h->pruned.as<unsigned long long>(), (long long)h->n_pruned, h->childmask.as<unsigned long long>(), h->count.as<int32_t>(), keep_lowdim);
                else hipLaunchKernelGGL(k_children_count<4>, dim3((unsigned)n), dim3(64), 0, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(),
                                        h->pruned.as<unsigned long long>(), (long long)h->n_pruned, h->childmask.as<unsigned long long>(), h->count.as<int32_t>(), keep_lowdim);
                // lean: children and their parent slots are sized by the bound n (n_c - k) and written without waiting for the count
                const double child_bound_bytes = (double)nn * std::max(h->n_c - k, 1) * (k + 2) * 4.0;
                children_lean = lean && child_bound_bytes <= 1.5e9;
                { int rcs = launch_scan(h, h->count.as<int32_t>(), h->offset.as<int32_t>(), n, children_lean ? dcnt + 20 : total); if (rcs) return rcs; }
                HIP_TRY(h, hipGetLastError());
                int32_t n_children = 0;
                if (children_lean) n_children = (int32_t)std::min<double>((double)nn * std::max(h->n_c - k, 1), 2147483647.0 / (k + 2));   // the bound, for the allocation only
                else { HIP_TRY(h, hipStreamSynchronize(st)); n_children = h->tot_host[0]; }
                h->n_children = children_lean ? 0 : n_children;
                if (n_children > 0) {
                    HIP_TRY(h, h->children.ensure((size_t)n_children * (k + 1) * sizeof(int32_t), st));
                    HIP_TRY(h, h->parent_slot_next.ensure((size_t)n_children * sizeof(int32_t), st));
                    hipLaunchKernelGGL(k_children_write, dim3((unsigned)n), dim3(64), 0, st, h->frontier.as<int32_t>(), n, k, h->mw,
                                       h->childmask.as<unsigned long long>(), h->offset.as<int32_t>(), h->children.as<int32_t>(),
                                       h->storing ? h->dict_stored[h->dict_cur].as<uint8_t>() : (const uint8_t *)nullptr, h->parent_slot_next.as<int32_t>());
                    HIP_TRY(h, hipGetLastError());
                }
            }
            hipLaunchKernelGGL(k_histogram, dim3(std::min(blocks256, 1024)), dim3(256), 0, st, h->status.as<uint8_t>(), n, ctr);
            HIP_TRY(h, hipGetLastError());
            if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[3], st));
            static_assert(sizeof(LevelCounters) % 4 == 0 && sizeof(LevelCounters) + 64 + 32 * 4 <= 4096, "LevelCounters + list lengths must fit the pinned block");
            hipLaunchKernelGGL(k_publish_words2, dim3(1), dim3(128), 0, st, reinterpret_cast<const unsigned int *>(ctr), (int)(sizeof(LevelCounters) / 4),
                               reinterpret_cast<const unsigned int *>(dcnt), 32, reinterpret_cast<unsigned int *>(h->tot_dev + 16));
            HIP_TRY(h, hipGetLastError());
            { int rcf = x1_flush(h); if (rcf) return rcf; }      // (MPC_X1_DEFER=2: k_x1 starts behind the level's last kernels)
            return MPC_OK;
        };
        // open parameter set: the candidates the verdict stage calls optimal are asked whether the reference's max-t LP is bounded (k_recession)
        auto recession = [&]() -> int {
            if (!h->theta_open || (flags & MPC_LEVEL_GRAPH)) return MPC_OK;
            hipLaunchKernelGGL(k_recession, dim3((unsigned)std::min<long long>(n, h->grid_v)), dim3(64), h->lds_v, st, h->Pv, h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>());
            HIP_TRY(h, hipGetLastError());
            return MPC_OK;
        };
        // verdict
        if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[0], st));
        if (h->fast && !h->force_v1) {
            const int32_t *fr = h->frontier.as<int32_t>();
            uint8_t *stp = h->status.as<uint8_t>();
            const DevProblem *pf = h->pf_dev.as<DevProblem>();
            // KKT solves + box screen, one thread per candidate (Schur/Cholesky mode, cardinality 1..8); the wave kernels
            // fetch the multipliers of the candidates the screen left open
            long long n_theta = n;
            const int32_t *theta_list = nullptr;
            bool kkt_listed = false;
            const int kd = k - h->targs.ne;   // rows the one-thread KKT kernel solves for (the equality rows are eliminated)
            if (h->kkt_mode == 0 && kd >= 1 && kd <= KKT_THREAD_MAX && h->no_kkt_thread != 1) {
                HIP_TRY(h, h->kkt_code.ensure(nn, st));
                HIP_TRY(h, h->kkt_L.ensure(nn * (size_t)k * (h->n_t + 1) * sizeof(double), st));
                kkc = h->kkt_code.as<uint8_t>(); kkl = h->kkt_L.as<double>();
                // (round 6) KKT_SPREAD lanes per candidate while that still leaves the level short of the chip's thread slots (k_kkt_thread's comment)
                const bool spread = h->kkt_spread > 0 && kd <= KKT_SPREAD_KMAX && n * KKT_SPREAD <= (long long)h->kkt_spread_threads;
                const dim3 g((unsigned)(spread ? (n * KKT_SPREAD + 255) / 256 : blocks256)), b(256);
                ThetaArgs ta = h->targs;
                // (round 5) a lean level lets the kernel list its own output: the theta stage's work list and the candidates its box
                // screen sends to the (x,theta) question -- two compactions of five launches each saved on the level's critical path
                const bool kkt_lists = lean && !h->no_kkt_lists && n <= 0x7fffffffLL;
                if (kkt_lists) {
                    HIP_TRY(h, h->theta_list.ensure(nn * sizeof(int32_t), st));
                    HIP_TRY(h, h->xq_list.ensure(nn * sizeof(int32_t), st));
                    ta.kt_list = h->theta_list.as<int32_t>(); ta.kt_n = dcnt + 0;
                    ta.kx_list = h->xq_list.as<int32_t>(); ta.kx_n = dcnt + 10;
                }
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[6], st));
#define MPC_LAUNCH_KKT_(K_, SP_) if (h->fast_t >= 4) hipLaunchKernelGGL((k_kkt_thread<K_, 10, SP_>), g, b, 0, st, pf, fr, n, h->kkt_code.as<uint8_t>(), h->kkt_L.as<double>(), stp, ta, ctr); \
                                    else if (h->fast_t >= 2) hipLaunchKernelGGL((k_kkt_thread<K_, 8, SP_>), g, b, 0, st, pf, fr, n, h->kkt_code.as<uint8_t>(), h->kkt_L.as<double>(), stp, ta, ctr); \
                                    else hipLaunchKernelGGL((k_kkt_thread<K_, 4, SP_>), g, b, 0, st, pf, fr, n, h->kkt_code.as<uint8_t>(), h->kkt_L.as<double>(), stp, ta, ctr)
#define MPC_LAUNCH_KKT(K_) case K_: MPC_LAUNCH_KKT_(K_, 1); break
#define MPC_LAUNCH_KKT_S(K_) case K_: if (spread) { MPC_LAUNCH_KKT_(K_, KKT_SPREAD); } else { MPC_LAUNCH_KKT_(K_, 1); } break
                switch (kd) { MPC_LAUNCH_KKT_S(1); MPC_LAUNCH_KKT_S(2); MPC_LAUNCH_KKT_S(3); MPC_LAUNCH_KKT_S(4); MPC_LAUNCH_KKT_S(5); MPC_LAUNCH_KKT_S(6); MPC_LAUNCH_KKT(7); MPC_LAUNCH_KKT(8); MPC_LAUNCH_KKT(9); MPC_LAUNCH_KKT(10); }
#undef MPC_LAUNCH_KKT_S
#undef MPC_LAUNCH_KKT
#undef MPC_LAUNCH_KKT_
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[7], st));
                kernel_timed[3] = true;
                HIP_TRY(h, hipGetLastError());
                int32_t n_todo = 0;
                if (kkt_lists) { theta_lean = true; n_todo = (int32_t)n; kkt_listed = true; }
                else {
                    if (lean) { int rcs = compact(ST_TODO, ST_TODO, nullptr, dcnt + 0); if (rcs) return rcs; theta_lean = true; n_todo = (int32_t)n; }
                    else { int rcs = compact(ST_TODO, ST_TODO, &n_todo); if (rcs) return rcs; }
                    HIP_TRY(h, h->theta_list.ensure(nn * sizeof(int32_t), st));
                    std::swap(h->theta_list, h->retry_list);   // compact() filled retry_list; keep it as the theta list
                }
                n_theta = n_todo;   // lean: the bound
                theta_list = h->theta_list.as<int32_t>();
            }
            // The quick test's thread pass BESIDE the theta stage (round 5).  On a level that keeps no dictionaries, k_kkt_thread's box
            // screen has already sent almost every candidate to the (x,theta) question (config 4's last level: 98.5 %), and that
            // question does not depend on the theta stage at all.  Their list is cut here and k_xq_thread takes it on the second
            // stream -- a kernel bound by the cache's request rate beside one bound by dependent fp64 latency -- ; the two meet
            // before the partition that follows the theta stage, which then sees only what the pass left open.
            // (the previous level's deferred k_x1 is joined by whoever reads dictionary records first: the thread pass on the second stream
            //  below -- the theta kernel reads none and goes ahead --, else the (x,theta) stage behind the theta stage)
            bool early_xq = false;
            if (kkc && lean && h->no_xq_early <= 0 && h->xq_thread != 0 && !h->no_xquick && !(flags & MPC_LEVEL_GRAPH) && h->have_prev_dict && h->have_parent_slot &&
                n >= h->xqt_min && !h->force_xqgroup && (h->no_xq_early < 0 || h->prev_regions < h->xq_early_regions || h->xq_early_regions <= 0)) {
                // (The pass then overlaps the theta stage AND the region stage: the partition behind the theta stage lists only the doubtful
                //  and the optimal candidates -- classes the pass never touches --, the region kernel starts on its stream, and the open
                //  candidates are listed when the pass has ended.  Config 4's last level: 1.86 ms with the pass behind the theta stage,
                //  1.72 beside it; config 3's: 3.45 / 3.25.  MPC_NO_XQ_EARLY=1: behind.)
                const int nxc_e = h->fast_x >= 2 ? 32 : 16;
                const long long sd_e = (long long)nxc_e * h->Pf.n_d0r, si_e = dict_ints(h->Pf.n_d0r, nxc_e, h->n_c);
                const bool will_store = gen_children && (double)nn * (sd_e * 8.0 + si_e * 4.0) / 1e9 <= h->dict_budget_gb;
                if (!will_store) {
                    if (!kkt_listed) {      // (else k_kkt_thread has listed them in xq_list, length in dcnt[10])
                        { int rcs = compact(ST_NEEDX, ST_NEEDX_SING, nullptr, dcnt + 10); if (rcs) return rcs; }
                        HIP_TRY(h, h->xq_list.ensure(nn * sizeof(int32_t), st));
                        std::swap(h->xq_list, h->retry_list);
                    }
                    DictCache dq{};
                    dq.stride_d = sd_e; dq.stride_i = si_e;
                    dq.parent_slot = h->parent_slot.as<int32_t>();
                    dq.prev_d = h->dict_d[1 - h->dict_cur].as<double>(); dq.prev_i = h->dict_i[1 - h->dict_cur].as<int32_t>();
                    dq.n_list_dev = dcnt + 10;
                    XqAlt alt{};
                    if ((h->xq_thread >= 2 || h->xq_thread < 0) && h->n_prev > 0 && h->n_prev <= 0x7fffffffLL && k >= 2 &&
                        h->children.cap >= (size_t)h->n_prev * (k - 1) * sizeof(int32_t) && h->dict_stored[1 - h->dict_cur].cap >= (size_t)h->n_prev) {
                        alt.prev_frontier = h->children.as<int32_t>(); alt.prev_stored = h->dict_stored[1 - h->dict_cur].as<uint8_t>();
                        alt.n_prev = (int)h->n_prev; alt.tries = h->xq_thread < 0 ? MPC_MAX_NC : h->xq_thread - 1;
                    }
                    HIP_TRY(h, hipEventRecord(h->ev_xfork, st));
                    HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_xfork, 0));
                    if (h->x1_pending) HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_x1done, 0));   // (the main stream joins later: x1_join before its own first reader)
                    const unsigned gt = (unsigned)std::min<long long>((n + 63) / 64, (long long)h->n_cu * h->xqt_wpc);
                    if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[10], h->stream2));
                    hipLaunchKernelGGL(k_xq_thread, dim3(gt), dim3(64), 0, h->stream2, pf, fr, k, h->xq_list.as<int32_t>(), (int)n, stp, ctr, dq, nxc_e, alt, XqPlan{});
                    if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[11], h->stream2));
                    HIP_TRY(h, hipGetLastError());
                    HIP_TRY(h, hipEventRecord(h->ev_xjoin, h->stream2));
                    xq_thread_timed = true;
                    early_xq = true; xq_early_ran = true;
                }
            }
            // Round 6: region wavefronts BESIDE the theta stage of a large last level (the queue form).  The theta
            // kernel appends every candidate it finds optimal to a queue (the level's opt_list) the moment it has decided it; a first
            // region launch of a few wavefronts per CU is already running on the side stream and builds regions while the theta stage is
            // still solving -- until now the region stage waited for the theta kernel's tail, a partition and a host read-back.  The
            // record buffers are sized by the length of the theta list, which k_kkt_thread has counted: published here, read by the host
            // while the theta kernel already runs.  A second launch behind the theta kernel drains the queue with the full width.
            if (r2_queue_ready && early_xq && kkt_listed && n_theta > 0 && !h->theta_open && !h->no_roverlap && h->test_late <= 0 && h->test_spare <= 0) {
                r2_early = true;
                if (h->r2_early_wpc > 0) {   // (the early launch's buffers are sized by the theta list's length: read while the theta kernel runs)
                    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned int *>(dcnt + 0), reinterpret_cast<unsigned int *>(h->tot_dev + 9), 1);
                    HIP_TRY(h, hipGetLastError());
                    HIP_TRY(h, hipEventRecord(h->ev_nth, st));
                }
            }
            if (n_theta > 0) {   // two-stage theta LP
                ThetaArgs ta = h->targs;
                if (r2_early) {
                    ta.optq = h->opt_list.as<int32_t>(); ta.q_tail = &ctr->q_tail;
                    // one theta wavefront per SIMD (the kernel is a tail of few long LPs: 0.40 / 0.33 ms with one / two, round 5) leaves
                    // every SIMD room for a region wavefront of the early launch; the theta wavefronts issue with priority (ThetaArgs::prio)
                    if (h->r2_early_wpc > 0) {
                        if (ta.wave_max > 0) ta.wave_max = std::min<int>(ta.wave_max, h->r2_early_thw * 4 * h->n_cu);
                        ta.prio = h->r2_early_prio;
                    }
                }
                ta.chunk = (int)std::max<long long>(1, std::min<long long>(16, n_theta / ((long long)h->grid_f * 8)));
                if (theta_lean) { ta.n_dev = dcnt + 0; ta.chunk = 0; }   // length and chunk rule on the device
                // wave slots the theta kernel takes (ThetaArgs::wave_div / wave_max): with the length on the device the kernel applies the
                // rule itself, here the host does
                long long grid_th = h->grid_f;
                if (ta.wave_max > 0) grid_th = std::min<long long>(grid_th, ta.wave_max);
                if (!theta_lean && ta.wave_div > 0) grid_th = std::min<long long>(grid_th, std::max<long long>(256, n_theta / ta.wave_div));
                if (!theta_lean) ta.chunk = (int)std::max<long long>(1, std::min<long long>(16, n_theta / (grid_th * 8)));
                const dim3 g((unsigned)std::min<long long>(theta_lean ? n_theta : (n_theta + ta.chunk - 1) / ta.chunk, grid_th)), b(64);
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[0], st));
                switch (h->fast_t) {
                    case 0: hipLaunchKernelGGL((k_theta2<4, 1>), g, b, h->lds_f, st, pf, fr, n_theta, k, stp, ctr, kkc, kkl, ta, theta_list); break;
                    case 1: hipLaunchKernelGGL((k_theta2<4, 2>), g, b, h->lds_f, st, pf, fr, n_theta, k, stp, ctr, kkc, kkl, ta, theta_list); break;
                    case 2: hipLaunchKernelGGL((k_theta2<8, 1>), g, b, h->lds_f, st, pf, fr, n_theta, k, stp, ctr, kkc, kkl, ta, theta_list); break;
                    case 3: hipLaunchKernelGGL((k_theta2<8, 2>), g, b, h->lds_f, st, pf, fr, n_theta, k, stp, ctr, kkc, kkl, ta, theta_list); break;
                    case 4: hipLaunchKernelGGL((k_theta2<10, 1>), g, b, h->lds_f, st, pf, fr, n_theta, k, stp, ctr, kkc, kkl, ta, theta_list); break;
                    default: hipLaunchKernelGGL((k_theta2<10, 2>), g, b, h->lds_f, st, pf, fr, n_theta, k, stp, ctr, kkc, kkl, ta, theta_list); break;
                }
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[1], st));
                kernel_timed[0] = true;
                n_theta_items = n_theta;
                HIP_TRY(h, hipGetLastError());
                if (r2_early) {
                    // the queue closes behind the theta kernel; its length and the number of doubtful candidates are published for the host
                    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(64), 0, st, &ctr->q_closed, 1u);
                    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, &ctr->q_tail, reinterpret_cast<unsigned int *>(h->tot_dev + 10), 1);
                    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, &ctr->n_retry_theta, reinterpret_cast<unsigned int *>(h->tot_dev + 11), 1);
                    HIP_TRY(h, hipGetLastError());
                    HIP_TRY(h, hipEventRecord(h->ev_part, st));
                    if (h->r2_early_wpc > 0) {
                        // ... and while the theta kernel runs: the early region launch (MPC_R2_EARLY_WPC > 0; off by default)
                        HIP_TRY(h, hipEventSynchronize(h->ev_nth));
                        const int32_t n_theta_host = h->tot_host[9];
                        const int32_t spare = (int32_t)std::min<long long>(std::max<long long>(2048, n_theta_host / 8), 1 << 20);
                        h->opt_ptr = h->opt_list.as<int32_t>();
                        int rcs = n_theta_host > 0 ? region2_launch(n_theta_host, spare, h->stream3, true, true) : MPC_OK;
                        if (rcs) return rcs;
                        r2_early_ntot = (long long)n_theta_host + spare;
                        r2_early_A = n_theta_host > 0;
                    }
                }
            }
            // ---- (x,theta) stage with the dictionary cache -------------------------------------------------------------
            const int nxc = h->fast_x >= 2 ? 32 : 16;
            h->dict_stride_d = (long long)nxc * h->Pf.n_d0r;   // column-major tableau
            h->dict_stride_i = dict_ints(h->Pf.n_d0r, nxc, h->n_c);
            DictCache dc{};
            dc.fresh_limit = h->x_fresh_limit; dc.second_max = h->x_second_max;
    dc.fresh_limit = h->x_fresh_limit; dc.second_max = h->x_second_max;
            dc.stride_d = h->dict_stride_d; dc.stride_i = h->dict_stride_i;
            if (h->have_prev_dict && h->have_parent_slot) {
                dc.parent_slot = h->parent_slot.as<int32_t>();
                dc.prev_d = h->dict_d[1 - h->dict_cur].as<double>();
                dc.prev_i = h->dict_i[1 - h->dict_cur].as<int32_t>();
            }
            h->storing = false;
            const double need_gb = (double)nn * (h->dict_stride_d * 8.0 + h->dict_stride_i * 4.0) / 1e9;
            if (gen_children && need_gb <= h->dict_budget_gb) {
                HIP_TRY(h, h->dict_d[h->dict_cur].ensure(nn * h->dict_stride_d * sizeof(double), st));
                HIP_TRY(h, h->dict_i[h->dict_cur].ensure(nn * h->dict_stride_i * sizeof(int32_t), st));
                HIP_TRY(h, h->dict_stored[h->dict_cur].ensure(nn, st));
                { int rcs = prep_zero(h->dict_stored[h->dict_cur].p, nn); if (rcs) return rcs; }
                dc.cur_d = h->dict_d[h->dict_cur].as<double>(); dc.cur_i = h->dict_i[h->dict_cur].as<int32_t>();
                dc.stored = h->dict_stored[h->dict_cur].as<uint8_t>();
                h->storing = true;
            }
            // The (x,theta) stage of a level that keeps dictionaries, in its one-step-plan form, needs no count from the host: its lists
            // come from the partition below, their lengths stay in device memory (dcnt[28..31]), its launches are sized by the bound n.
            // Round 5 (`x_first`): it is queued BEHIND THE PARTITION AT ONCE, and only then does the host wait for the counts that size
            // the region stage -- until then the stage waited ~0.1 ms per large level for that read-back and the region launch's host work
            // (tools/timeline.sh: 110 us between the partition and the plan pass on level 4 of config 4).
            bool x_done = false;
            auto launch_plans = [&](const int32_t *needx_list_, int n_needx_, long long n_bound, const int32_t *n_needx_dev, const int32_t *n_pre1_dev, const int32_t *n_pre2_dev) -> int {
                // One-step plans (round 5).  Every candidate that needs a dictionary is asked by ONE THREAD whether a parent's record is
                // one known step away (k_xq_thread in plan mode: the generating parent, then the candidate's other parents); k_x1 then
                // streams that record through the step -- no tableau in registers, no pricing, no ratio test --, and only what has no
                // such plan goes through the register simplex k_x2 as before, from its generating parent.
                { int rcq = prep_flush(); if (rcq) return rcq; }
                HIP_TRY(h, h->x1_buf.ensure((6 * nn + 16) * sizeof(int32_t), st));
                int32_t *xb = h->x1_buf.as<int32_t>();
                XqPlan pl{};
                pl.plan_slot = xb; pl.plan_step = xb + nn; pl.x1_list = xb + 2 * nn; pl.x1_n = dcnt + 12;
                for (int sg = 0; sg < 3; ++sg) { pl.rest[sg] = xb + (3 + sg) * nn; pl.rest_n[sg] = dcnt + 13 + sg; }
                pl.pre1 = dc.pre1; pl.n_pre1 = dc.n_pre1; pl.pre2 = dc.pre2; pl.n_pre2 = dc.n_pre2;
                pl.n_pre1_dev = n_pre1_dev; pl.n_pre2_dev = n_pre2_dev;
                XqAlt alt{};
                if (h->x1 >= 2 && h->n_prev > 0 && h->n_prev <= 0x7fffffffLL && k >= 2 &&
                    h->children.cap >= (size_t)h->n_prev * (k - 1) * sizeof(int32_t) && h->dict_stored[1 - h->dict_cur].cap >= (size_t)h->n_prev) {
                    alt.prev_frontier = h->children.as<int32_t>(); alt.prev_stored = h->dict_stored[1 - h->dict_cur].as<uint8_t>();
                    alt.n_prev = (int)h->n_prev; alt.tries = MPC_MAX_NC;
                }
                DictCache dq = dc;
                dq.n_list_dev = n_needx_dev;
                const DevProblem *pfx = h->pf_dev.as<DevProblem>();
                const int32_t *frx = h->frontier.as<int32_t>();
                uint8_t *stx = h->status.as<uint8_t>();
                const unsigned gt = (unsigned)std::min<long long>((n_bound + 63) / 64, (long long)h->n_cu * h->xqt_wpc);
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[10], st));
                hipLaunchKernelGGL(k_xq_thread, dim3(gt), dim3(64), 0, st, pfx, frx, k, needx_list_, n_needx_, stx, ctr, dq, nxc, alt, pl);
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[11], st));
                xq_thread_timed = true;
                const unsigned g1 = (unsigned)std::min<long long>(n_bound, (long long)h->n_cu * h->x1_wpc);
                // Round 6: the stream of one-step dictionaries goes to stream4 and is NOT waited for by this level: its records are read
                // by the next level's (x,theta) kernels only (x1_join), the plan pass has already marked the planned candidates as
                // "dictionary stored" for k_children_write.  The end of this level (k_x2 for the unplanned rest, children, counters), the
                // hand-over and the next level's KKT kernel run beside it.
                const bool defer = h->x1_defer > 0 && !h->timing && h->stream4;
                const bool late = defer && h->x1_defer >= 2;      // issued by x1_flush behind the level's last kernels
                hipStream_t sx1 = defer ? h->stream4 : st;
                if (defer && !late) { HIP_TRY(h, hipEventRecord(h->ev_x1go, st)); HIP_TRY(h, hipStreamWaitEvent(sx1, h->ev_x1go, 0)); }
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[12], st));
                LevelCounters *ctr_x1 = defer ? (LevelCounters *)nullptr : ctr;   // (its pivot count would land in the next level's counters)
                // MPC_X1_LDS_CAP = c > 0: the deferred kernel asks for (144 KB / c) of LDS it never touches, which holds it to c wavefronts per compute unit
                const unsigned lds_x1 = defer && h->x1_lds_cap > 0 ? (unsigned)((144 * 1024) / h->x1_lds_cap) & ~255u : 0u;
                const int32_t *x1_n_ptr = pl.x1_n;
                if (late) {
                    // the list's length lives among the level's counters, which the next level clears: the late launch reads a copy
                    int32_t *keep = xb + 6 * nn;
                    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned int *>(pl.x1_n), reinterpret_cast<unsigned int *>(keep), 1);
                    x1_n_ptr = keep;
                }
                const bool two = (h->fast_x & 1) != 0;
                const int32_t *x1_list_p = pl.x1_list, *plan_slot_p = pl.plan_slot, *plan_step_p = pl.plan_step;
                auto launch_x1 = [h, two, g1, lds_x1, sx1, pfx, x1_list_p, x1_n_ptr, ctr_x1, dc, nxc, plan_slot_p, plan_step_p, late, defer]() -> int {
                    if (late) { HIP_TRY(h, hipEventRecord(h->ev_x1go, h->stream)); HIP_TRY(h, hipStreamWaitEvent(sx1, h->ev_x1go, 0)); }
                    if (two) hipLaunchKernelGGL((k_x1<2>), dim3(g1), dim3(64), lds_x1, sx1, pfx, x1_list_p, x1_n_ptr, ctr_x1, dc, nxc, plan_slot_p, plan_step_p);
                    else hipLaunchKernelGGL((k_x1<1>), dim3(g1), dim3(64), lds_x1, sx1, pfx, x1_list_p, x1_n_ptr, ctr_x1, dc, nxc, plan_slot_p, plan_step_p);
                    HIP_TRY(h, hipGetLastError());
                    if (defer) { HIP_TRY(h, hipEventRecord(h->ev_x1done, sx1)); h->x1_pending = true; }
                    return MPC_OK;
                };
                if (late) h->x1_stash = launch_x1;
                else { int rcx = launch_x1(); if (rcx) return rcx; }
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[13], st));
                // what is left: the register simplex, list lengths on the device
                DictCache dr = dc;
                dr.pre1 = pl.rest[0]; dr.n_pre1 = 0; dr.n_pre1_dev = pl.rest_n[0];
                dr.pre2 = pl.rest[1]; dr.n_pre2 = 0; dr.n_pre2_dev = pl.rest_n[1];
                dr.n_list_dev = pl.rest_n[2]; dr.chunk = 0;
                const long long grid_r = (long long)h->n_cu * std::min<long long>(4, h->x2_wpc);
                const dim3 gg((unsigned)std::max<long long>(1, std::min<long long>(n_bound, grid_r))), bb(64);
                switch (h->fast_x) {
                    case 0: hipLaunchKernelGGL((k_x2<16, 1>), gg, bb, 0, st, pfx, frx, k, pl.rest[2], n_needx_, stx, ctr, dr); break;
                    case 1: hipLaunchKernelGGL((k_x2<16, 2>), gg, bb, 0, st, pfx, frx, k, pl.rest[2], n_needx_, stx, ctr, dr); break;
                    case 2: hipLaunchKernelGGL((k_x2<32, 1>), gg, bb, 0, st, pfx, frx, k, pl.rest[2], n_needx_, stx, ctr, dr); break;
                    default: hipLaunchKernelGGL((k_x2<32, 2>), gg, bb, 0, st, pfx, frx, k, pl.rest[2], n_needx_, stx, ctr, dr); break;
                }
                HIP_TRY(h, hipGetLastError());
                x1_ran = true;
                return MPC_OK;
            };
            // (behind the theta kernel: everything the main stream and the side streams launch from here on may read dictionary records)
            { int rcj = x1_join(h); if (rcj) return rcj; }
            // One partition after the theta stage: [0] numerically doubtful (status 7), [1] feasible and [2] optimal (decided
            // in theta space; they only need a dictionary for their children), [3] feasibility still open.
            int32_t cntA[PART_CLASSES] = {0, 0, 0, 0};
            const bool x_first = lean && !early_xq && h->x_first && h->storing && dc.parent_slot && h->x1 > 0 && n >= std::max<long long>(h->x1_min, h->x_first_min) && n <= h->x_first_max &&
                                 n <= 0x7fffffffLL && !(flags & MPC_LEVEL_GRAPH);
            if (x_first) {
                { int rcs = partition({{ST_RETRY, 0}, {ST_FEASIBLE, 1}, {ST_OPT_PENDING, 2}, {ST_NEEDX, 3}, {ST_NEEDX_SING, 3}}, cntA, true, dcnt + 28); if (rcs) return rcs; }
                hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, reinterpret_cast<const unsigned int *>(dcnt + 28), reinterpret_cast<unsigned int *>(h->tot_dev + 12), PART_CLASSES);
                HIP_TRY(h, hipGetLastError());
                HIP_TRY(h, hipEventRecord(h->ev_part, st));
                dc.pre1 = part_list(1); dc.pre2 = part_list(2);
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[2], st));
                { int rcs = launch_plans(part_list(3), (int)n, n, dcnt + 31, dcnt + 29, dcnt + 30); if (rcs) return rcs; }
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[3], st));
                kernel_timed[1] = true;
                x_done = true;
                HIP_TRY(h, hipEventSynchronize(h->ev_part));      // the counts that size the region stage (and the doubtful candidates' side stream)
                for (int c = 0; c < PART_CLASSES; ++c) cntA[c] = h->tot_host[12 + c];
            } else if (early_xq) {
                // The thread pass is still rewriting the statuses of ITS candidates (NEEDX -> feasible / infeasible / singular) on the second
                // stream: this partition asks only for the two classes it never touches -- doubtful and optimal candidates of the theta
                // stage --, so that the region stage can start beside it; the open candidates are listed when the pass has ended (below).
                if (r2_early) {
                    // (queue form: the optimal candidates are the queue; region wavefronts are already rewriting their statuses.  The doubtful
                    //  candidates are counted by the theta kernel itself: without one -- the usual case -- no partition is needed at all, and
                    //  its 1024-thread workgroups would wait for room beside the region wavefronts: 0.75 ms on config 4's last level)
                    HIP_TRY(h, hipEventSynchronize(h->ev_part));
                    if (h->tot_host[11] > 0) { int rcs = partition({{ST_RETRY, 0}}, cntA); if (rcs) return rcs; }
                    cntA[2] = h->tot_host[10];
                    h->opt_ptr = h->opt_list.as<int32_t>();
                } else { int rcs = partition({{ST_RETRY, 0}, {ST_OPT_PENDING, 2}}, cntA); if (rcs) return rcs; }
                cntA[3] = (int32_t)std::min<long long>(n, 0x7fffffffLL);     // a bound, for the decisions that follow; the count comes after the join
            } else { int rcs = partition({{ST_RETRY, 0}, {ST_FEASIBLE, 1}, {ST_OPT_PENDING, 2}, {ST_NEEDX, 3}, {ST_NEEDX_SING, 3}}, cntA); if (rcs) return rcs; }
            // The doubtful candidates are re-solved by the LDS engine, which can refactorise its basis: a few hundred
            // long-running wavefronts.  They run on the side stream while the (x,theta) stage fills the GPU; their results are
            // applied to the status array after the join.
            const int32_t n_early = cntA[0];
            if (n_early > 0) {
                HIP_TRY(h, h->vretry_list.ensure(nn * sizeof(int32_t), st));
                HIP_TRY(h, h->status_tmp.ensure(nn, st));
                if (x_done) {   // (the main stream already carries the (x,theta) stage: the side stream starts behind the partition's event)
                    HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_part, 0));
                    HIP_TRY(h, hipMemcpyAsync(h->vretry_list.p, part_list(0), (size_t)n_early * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream2));
                } else {
                    HIP_TRY(h, hipMemcpyAsync(h->vretry_list.p, part_list(0), (size_t)n_early * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
                    HIP_TRY(h, hipEventRecord(h->ev_fork, st));
                    HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                }
                hipLaunchKernelGGL(k_verdict, dim3((unsigned)std::min<long long>(n_early, h->grid_v)), dim3(64), h->lds_v, h->stream2, h->Pv,
                                   h->frontier.as<int32_t>(), (long long)n_early, k, h->status_tmp.as<uint8_t>(), ctr, h->vretry_list.as<int32_t>(), (const int32_t *)nullptr);
                HIP_TRY(h, hipGetLastError());
                HIP_TRY(h, hipEventRecord(h->ev_join, h->stream2));
            }
            // The region stage needs the theta stage's verdicts only (no later stage turns a candidate optimal, except the
            // re-solved doubtful ones): it starts now on its own stream and runs under the (x,theta) stage -- a few thousand
            // long wavefronts at two per SIMD whose tail the streaming kernels of the (x,theta) stage fill.
            int32_t region_extra = 0;
            // x_items: what the (x,theta) stage has to do.  Below roverlap_min items there is nothing to hide the region stage
            // under; from roverlap_long items on the stage outlasts the region kernel anyway, which then runs one wavefront per
            // candidate (splitting a candidate over four wavefronts shortens its latency but takes registers from four SIMDs).
            const long long x_items = (long long)cntA[3] + (h->storing ? (long long)cntA[1] + cntA[2] : 0);
            // (the grouped quick test keeps four 128-register wavefronts per SIMD busy through LDS latency: one region wavefront on
            // a CU halves that CU's share of it -- config 3's last level loses 0.1-0.25 ms with the region stage under it)
            const size_t lds_q = (size_t)h->dict_stride_d * sizeof(double) + (size_t)h->dict_stride_i * sizeof(int32_t);
            const bool quick_test = cntA[3] > 0 && !(flags & MPC_LEVEL_GRAPH) && !h->storing && dc.parent_slot && !h->no_xquick;
            const bool use_grouped = quick_test && !h->no_xqgroup && ((h->last_level_n > 0 && (long long)cntA[3] >= 10 * h->last_level_n) || h->force_xqgroup) &&
                                     cntA[3] >= 4096 && lds_q <= 64 * 1024 && !early_xq;   // (after the thread pass few candidates per parent are left)
            if (r2_early_A) {
                // queue form: the early launch is running; the queue is closed and holds cntA[2] candidates -- the drain launch takes what is
                // left with the full width, and the host learns how many chunks of slots there are
                h->n_opt = cntA[2];
                region_extra = (int32_t)std::max<long long>(0, r2_early_ntot - cntA[2]);     // every slot behind the queue's is a spare one
                if (h->so.active) {
                    h->so.n_chunks = (int)(((long long)cntA[2] + (1ll << h->so.shift) - 1) >> h->so.shift);
                    h->cw_chunks = h->so.n_chunks;
                }
                if (cntA[2] > 0) { int rcs = region2_drain(cntA[2]); if (rcs) return rcs; r2_drained = true; }
                region_launched = true;
                stream_ready(h);
            } else
            if (!h->no_roverlap && !h->theta_open && cntA[2] > 0 && h->fast_r >= 0 && !(flags & MPC_LEVEL_GRAPH) && x_items >= h->roverlap_min && (!use_grouped || h->xqg_overlap)) {
                // Candidates that turn out optimal later -- re-solved doubtful ones: the n_early of the theta stage, rarely one of
                // the (x,theta) stage -- get spare slots behind the launch's and take the LDS-engine route of the candidates
                // k_region2 gives up on.  The spare slots cover every re-solved candidate of the theta stage plus up to 1,024 of the
                // (x,theta) stage (candidates whose theta stage found no feasible parameter and whose re-solve, in other arithmetic,
                // calls them optimal: none has been seen).  A candidate is never demoted for want of a slot: late candidates beyond
                // the spare slots make the level fail with MPC_ERR_CAPACITY, and the driver repeats the solve with the region stage
                // behind the (x,theta) stage (mpc_set_region_overlap(h, 0)), where every optimal candidate is known at launch.
                const int32_t hold = std::min<int32_t>(std::max(h->test_late, 0), cntA[2] - 1);
                const int32_t n_launch = cntA[2] - hold;
                region_extra = std::max(0, n_early + hold + std::min<int32_t>(cntA[3], 1024) - std::max(h->test_spare, 0));
                if (!r2_early) { rprep.copy_dst = h->opt_list.as<int32_t>(); rprep.copy_src = part_list(2); rprep.copy_n = n_launch; rprep_any = true; }   // (queue form: the list IS the queue)
                h->opt_ptr = h->opt_list.as<int32_t>();
                h->n_opt = n_launch;
                int rcs = region2_launch(n_launch, region_extra, h->stream3, x_items >= h->roverlap_long);
                if (rcs) return rcs;
                region_launched = true;
            }
            auto launch_x = [&](const int32_t *ls, int n_items, const DictCache &d0) -> int {
                // ctr->work_x is zero: the counters were cleared at the start of the level and this is the level's only k_x2 launch
                { int rcs = prep_flush(); if (rcs) return rcs; }
                DictCache d = d0;
                // wavefronts per CU of the persistent launch: fewer for fewer items (an item is faster the fewer wavefronts share its SIMD; round 4,
                // config 4: level 3, 15.7 k items, 0.445 ms with 16 per CU, 0.33 with 4; level 4, 138 k items, 1.134 / 1.083 / 1.161 ms with 16 / 12 / 8)
                const long long n_all = (long long)n_items + d.n_pre1 + d.n_pre2;
                const long long wpc = std::max<long long>(std::min<long long>(4, h->x2_wpc), std::min<long long>(h->x2_wpc, n_all / ((long long)h->x2_div * h->n_cu)));
                const long long grid_x = (long long)h->n_cu * wpc;
                d.chunk = (int)std::max<long long>(1, std::min<long long>(16, n_all / (grid_x * 8)));
                if (xq_lean) { d.n_list_dev = dcnt + 8; d.chunk = 0; }   // length of `ls` and chunk rule on the device
                const dim3 gg((unsigned)std::min<long long>(xq_lean ? n_all : (n_all + d.chunk - 1) / d.chunk, grid_x)), bb(64);
                switch (h->fast_x) {
                    case 0: hipLaunchKernelGGL((k_x2<16, 1>), gg, bb, 0, st, pf, fr, k, ls, n_items, stp, ctr, d); break;
                    case 1: hipLaunchKernelGGL((k_x2<16, 2>), gg, bb, 0, st, pf, fr, k, ls, n_items, stp, ctr, d); break;
                    case 2: hipLaunchKernelGGL((k_x2<32, 1>), gg, bb, 0, st, pf, fr, k, ls, n_items, stp, ctr, d); break;
                    default: hipLaunchKernelGGL((k_x2<32, 2>), gg, bb, 0, st, pf, fr, k, ls, n_items, stp, ctr, d); break;
                }
                HIP_TRY(h, hipGetLastError());
                return MPC_OK;
            };
            if (h->storing) {
                // candidates the theta stage already decided (feasible / optimal) expand too: their children get a
                // dictionary to start from (status untouched).  They ride in front of the open candidates in the same launch.
                dc.pre1 = part_list(1); dc.n_pre1 = cntA[1];
                dc.pre2 = part_list(2); dc.n_pre2 = cntA[2];
            }
            if (early_xq) {
                // the thread pass has ended: what it left open (and what the theta stage left open) is listed now
                HIP_TRY(h, hipStreamWaitEvent(st, h->ev_xjoin, 0));
                int32_t n_left = 0;
                { int rcs = compact(ST_NEEDX, ST_NEEDX_SING, &n_left); if (rcs) return rcs; }
                std::swap(h->xq_list, h->retry_list);     // (xq_list held the pass's input: free again)
                cntA[3] = n_left;
            }
            int32_t n_needx = cntA[3];
            const int32_t *needx_list = early_xq ? h->xq_list.as<int32_t>() : part_list(3);
            h->n_needx = n_needx;
            if ((flags & MPC_LEVEL_GRAPH) && n_needx > 0) {
                // connected-graph traversal: only "is the critical region non-empty" is asked; the candidates whose theta stage
                // left feasibility open simply have no region
                hipLaunchKernelGGL(k_close_open, dim3((unsigned)((n_needx + 255) / 256)), dim3(256), 0, st, needx_list, (int)n_needx, stp);
                HIP_TRY(h, hipGetLastError());
                n_needx = 0;
            }
            bool xretry_forked = false;
            if (quick_test) {
                // last level: decisions only -- the quick test on three vectors of the parent's dictionary first
                DictCache dq = dc;
                const long long grid_q = (long long)h->n_cu * h->xq_wpc;
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[8], st));
                n_xq_items = n_needx;
                // First pass, one THREAD per candidate (k_xq_thread, round 5): whatever the first ratio test of the hinted column
                // decides -- against the generating parent's record, then against the records of the candidate's OTHER parents (the
                // previous level's sets that lack one of its other members).  What stays open is compacted (length on the device) and
                // goes to the wavefront kernel unchanged.  Against the generating parent alone the pass decides 67 % of config 4's
                // last level and gains nothing (the wavefront kernel's time is the other third, candidates that need two to sixteen
                // dependent pivots from THAT vertex: level 5 2.36 ms with, 2.35 without); from another parent's vertex almost all
                // of those are one ratio test away too: 99.76 % decided, level 5 2.35 -> 1.75 ms; config 3 93 %, 4.75 -> 3.43 ms.
                const int32_t *xq_list = needx_list;
                int32_t xq_n = n_needx;
                bool xqt_lean = false;
                if (h->xq_thread != 0 && n_needx >= h->xqt_min && !early_xq) {
                    const unsigned gt = (unsigned)std::min<long long>(((long long)n_needx + 63) / 64, (long long)h->n_cu * h->xqt_wpc);
                    if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[10], st));
                    XqAlt alt{};
                    if ((h->xq_thread >= 2 || h->xq_thread < 0) && h->n_prev > 0 && h->n_prev <= 0x7fffffffLL && h->have_prev_dict && k >= 2 &&
                        h->children.cap >= (size_t)h->n_prev * (k - 1) * sizeof(int32_t) && h->dict_stored[1 - h->dict_cur].cap >= (size_t)h->n_prev) {
                        alt.prev_frontier = h->children.as<int32_t>(); alt.prev_stored = h->dict_stored[1 - h->dict_cur].as<uint8_t>();
                        alt.n_prev = (int)h->n_prev; alt.tries = h->xq_thread < 0 ? MPC_MAX_NC : h->xq_thread - 1;   // (the kernel stops at the candidate's inequality members)
                    }
                    hipLaunchKernelGGL(k_xq_thread, dim3(gt), dim3(64), 0, st, pf, fr, k, needx_list, n_needx, stp, ctr, dq, nxc, alt, XqPlan{});
                    if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[11], st));
                    HIP_TRY(h, hipGetLastError());
                    xq_thread_timed = true;
                    if (lean && !use_grouped) { int rcs = compact(ST_NEEDX, ST_NEEDX_SING, nullptr, dcnt + 9); if (rcs) return rcs; xqt_lean = true; }   // (the grouped form sizes its group scan on the host)
                    else { int32_t n_left = 0; int rcs = compact(ST_NEEDX, ST_NEEDX_SING, &n_left); if (rcs) return rcs; xq_n = n_left; }
                    HIP_TRY(h, h->xq_list.ensure(nn * sizeof(int32_t), st));
                    std::swap(h->xq_list, h->retry_list);   // compact() filled retry_list; keep it as the wavefront kernel's list
                    xq_list = h->xq_list.as<int32_t>();
                }
                dq.chunk = (int)std::max<long long>(1, std::min<long long>(16, xq_n / (grid_q * 4)));
                if (xqt_lean) { dq.n_list_dev = dcnt + 9; dq.chunk = 0; dq.skip_below = (h->fast_x & 1) ? 0 : h->xq_skip_below; }   // length and chunk rule on the device; a short list is left to k_x2
                // doubtful pivots are flagged by the quick test itself and re-solved at once on the second stream (below), unless that stream
                // is busy with the theta stage's own doubtful candidates (their statuses are still ST_RETRY in the status array)
                const bool xq_flags_retry = h->xq_retry && n_early == 0 && lean;
                dq.flag_retry = xq_flags_retry ? 1 : 0;
                const dim3 gg((unsigned)std::max<long long>(1, std::min<long long>(xqt_lean ? (long long)xq_n : ((long long)xq_n + dq.chunk - 1) / dq.chunk, grid_q))), bb(64);
                // Grouped by parent when a parent has many open children (config 3: 12.6 per parent, -0.5 ms; config 4: 7.1 per
                // parent, where the per-candidate reads of k_xq are cheaper than one 16 KB copy per parent, +0.45 ms): threshold 10.
                if (use_grouped && xq_n > 0) {
                    // grouped by parent: the record is staged in LDS once per parent (k_xq_grouped)
                    HIP_TRY(h, h->xq_groups.ensure((size_t)xq_n * sizeof(int32_t), st));
                    const int nbq = (xq_n + 255) / 256;
                    hipLaunchKernelGGL(k_group_flags, dim3(nbq), dim3(256), 0, st, xq_list, xq_n, dq.parent_slot, h->flag.as<int32_t>());
                    { int rcs = launch_scan(h, h->flag.as<int32_t>(), h->pos.as<int32_t>(), xq_n, h->scratch.as<int32_t>()); if (rcs) return rcs; }
                    hipLaunchKernelGGL(k_scatter_index, dim3(nbq), dim3(256), 0, st, h->flag.as<int32_t>(), h->pos.as<int32_t>(), (long long)xq_n, h->xq_groups.as<int32_t>());
                    const int per_cu = std::max(1, std::min(h->xqg_per_cu, (int)((160 * 1024) / (lds_q + 64))));
                    const dim3 gq((unsigned)std::min<long long>(xq_n, (long long)h->n_cu * per_cu)), bq(256);
                    if (lds_q > 48 * 1024) {
                        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_xq_grouped<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
                        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_xq_grouped<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q));
                    }
                    if (h->fast_x & 1) hipLaunchKernelGGL((k_xq_grouped<2>), gq, bq, lds_q, st, pf, fr, k, xq_list, xq_n, stp, ctr, dq, nxc, h->xq_groups.as<int32_t>(), h->scratch.as<int32_t>());
                    else hipLaunchKernelGGL((k_xq_grouped<1>), gq, bq, lds_q, st, pf, fr, k, xq_list, xq_n, stp, ctr, dq, nxc, h->xq_groups.as<int32_t>(), h->scratch.as<int32_t>());
                } else if (xq_n == 0) { /* the first pass decided everything (host-known length) */ }
                else if (!xqt_lean && !(h->fast_x & 1) && xq_n < h->xq_skip_below) { /* a short list (host-known length) is left to k_x2: k_xq's comment */ }
                else if (h->fast_x & 1) hipLaunchKernelGGL((k_xq<2>), gg, bb, 0, st, pf, fr, k, xq_list, xq_n, stp, ctr, dq, nxc);
                else hipLaunchKernelGGL((k_xq<1>), gg, bb, 0, st, pf, fr, k, xq_list, xq_n, stp, ctr, dq, nxc);
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[9], st));
                kernel_timed[4] = true;
                HIP_TRY(h, hipGetLastError());
                if (xq_flags_retry) {
                    // The candidates the quick test flagged as doubtful (config 3's last level: 217) are re-solved by the LDS engine NOW, on the
                    // second stream, beside k_x2 and the end of the level -- rounds 1-4 let k_x2 repeat their runs to flag them, and started
                    // the re-solve (0.6 ms of a few hundred long wavefronts) behind everything else, then repeated the end of the level.
                    { int rcs = compact(ST_RETRY, ST_RETRY, nullptr, dcnt + 11); if (rcs) return rcs; }
                    HIP_TRY(h, h->xretry_list.ensure(nn * sizeof(int32_t), st));
                    std::swap(h->xretry_list, h->retry_list);
                    HIP_TRY(h, hipEventRecord(h->ev_xfork, st));
                    HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_xfork, 0));
                    HIP_TRY(h, hipMemsetAsync(&ctr->work_retry, 0, sizeof(unsigned int), h->stream2));
                    hipLaunchKernelGGL(k_verdict, dim3((unsigned)std::max<long long>(1, std::min<long long>(std::min<long long>(xq_n, 4096), h->grid_v))), dim3(64), h->lds_v, h->stream2, h->Pv,
                                       h->frontier.as<int32_t>(), (long long)xq_n, k, stp, ctr, h->xretry_list.as<int32_t>(), dcnt + 11);
                    HIP_TRY(h, hipGetLastError());
                    HIP_TRY(h, hipEventRecord(h->ev_xjoin, h->stream2));
                    xretry_forked = true;
                }
                int32_t n_left = 0;
                if (lean) { int rcs = compact(ST_NEEDX, ST_NEEDX_SING, nullptr, dcnt + 8); if (rcs) return rcs; xq_lean = true; n_left = n_needx; }
                else { int rcs = compact(ST_NEEDX, ST_NEEDX_SING, &n_left); if (rcs) return rcs; }
                n_needx = n_left;   // lean: the bound (everything the quick test was given)
                needx_list = h->retry_list.as<int32_t>();
            }
            if (x_done) n_x_items = n_needx + dc.n_pre1 + dc.n_pre2;      // (queued behind the partition, above)
            else if (n_needx + dc.n_pre1 + dc.n_pre2 > 0) {   // feasibility for the candidates left open (+ dictionary-only items)
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[2], st));
                int rcs = MPC_OK;
                const long long n_dict = (long long)n_needx + dc.n_pre1 + dc.n_pre2;
                if (h->storing && dc.parent_slot && h->x1 > 0 && !xq_lean && n_dict >= h->x1_min) {
                    rcs = launch_plans(needx_list, n_needx, n_dict, nullptr, nullptr, nullptr);
                } else rcs = launch_x(needx_list, n_needx, dc);
                if (rcs) return rcs;
                if (h->timing) HIP_TRY(h, hipEventRecord(h->kev[3], st));
                kernel_timed[1] = true;
                n_x_items = n_needx + dc.n_pre1 + dc.n_pre2;
            }
            // A storing level without (x,theta) items has not flushed `prep` (launch_x does): the clear of dict_stored[dict_cur] queued in it
            // must still run before k_children_write reads that array (stale "stored" bytes of two levels ago would send a child
            // to another candidate's dictionary).
            { int rcs = prep_flush(); if (rcs) return rcs; }
            // doubtful candidates of the (x,theta) stage (rare) take the same route on the main stream
            int32_t n_retry = 0;
            if (n_early > 0) {
                // the early retries still carry status 7: mark them so that this compaction does not pick them up again
                HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
                hipLaunchKernelGGL(k_apply_status, dim3((unsigned)((n_early + 255) / 256)), dim3(256), 0, st, h->vretry_list.as<int32_t>(), (int)n_early,
                                   h->status_tmp.as<uint8_t>(), stp);
                HIP_TRY(h, hipGetLastError());
                HIP_TRY(h, hipMemsetAsync(&ctr->work_retry, 0, sizeof(unsigned int), st));
            }
            // second partition: [0] doubtful candidates of the (x,theta) stage, [2] the optimal candidates for the region stage
            int32_t cntB[PART_CLASSES] = {0, 0, 0, 0};
            if (xretry_forked) {   // the re-solved doubtful candidates of the quick test carry their final statuses now
                HIP_TRY(h, hipStreamWaitEvent(st, h->ev_xjoin, 0));
                HIP_TRY(h, hipMemsetAsync(&ctr->work_retry, 0, sizeof(unsigned int), st));
            }
            if (region_launched) HIP_TRY(h, hipStreamWaitEvent(st, h->ev_rjoin, 0));   // the region kernel rewrites statuses
            if (r2_drained) HIP_TRY(h, hipStreamWaitEvent(st, h->ev_rjoin2, 0));
            bool have_cntB = false;
            if (region_launched && lean && !h->no_spec_tail) {
                // The overlapped region stage has finished (the stream waits for it above), so in the usual case nothing is left to do but
                // the end of the level: no doubtful candidate of the (x,theta) stage, no late optimal one, none that k_region2 gave up on.
                // The three counts that say so used to cost three host round trips (this partition, the n_rretry read-back, the final one:
                // 45-70 us of idle device per large level, tools/timeline.sh); now the end of the level is queued right behind the
                // partition, ONE synchronisation reads everything, and only when a count is not zero the classic sequence below runs and
                // the end of the level is repeated (its two accumulating counters are cleared; everything else it wrote is overwritten).
                { int rcs = partition({{ST_RETRY, 0}, {ST_OPT_PENDING, 2}}, cntB, true); if (rcs) return rcs; }
                hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, &ctr->n_rretry, reinterpret_cast<unsigned int *>(h->tot_dev + 8), 1);
                if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], st));
                if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], st));
                { int rcs = queue_tail(); if (rcs) return rcs; }
                HIP_TRY(h, hipStreamSynchronize(st));
                for (int c = 0; c < PART_CLASSES; ++c) cntB[c] = h->tot_host[12 + c];
                have_cntB = true;
                if (cntB[0] == 0 && cntB[2] == 0 && h->tot_host[8] == 0) tail_done = true;
                else {
                    HIP_TRY(h, hipMemsetAsync(&ctr->n_pruned_new, 0, sizeof(unsigned int), st));
                    HIP_TRY(h, hipMemsetAsync(ctr->status, 0, sizeof(ctr->status), st));
                    h->n_children = 0;
                }
            }
            { int rcs = recession(); if (rcs) return rcs; }
            if (!have_cntB) { int rcs = partition({{ST_RETRY, 0}, {ST_OPT_PENDING, 2}}, cntB); if (rcs) return rcs; }
            n_retry = cntB[0];
            if (n_retry > 0 && !region_launched && !h->no_roverlap && !h->theta_open && cntB[2] > 0 && h->fast_r >= 0 && !(flags & MPC_LEVEL_GRAPH)) {
                // the region stage was not started under the (x,theta) stage (grouped quick test, short stage): it runs beside the
                // re-solve of the doubtful candidates instead -- two small, latency-bound kernels; what the re-solve finds optimal
                // takes the spare slots (config 3's last level: 235 re-solved candidates, 0.4 ms)
                region_extra = n_retry;
                rprep.copy_dst = h->opt_list.as<int32_t>(); rprep.copy_src = part_list(2); rprep.copy_n = cntB[2]; rprep_any = true;   // (on the region kernel's stream)
                h->opt_ptr = h->opt_list.as<int32_t>();
                h->n_opt = cntB[2];
                int rcs = region2_launch(cntB[2], region_extra, h->stream3, false);
                if (rcs) return rcs;
                region_launched = true;
            }
            if (n_retry > 0) {
                hipLaunchKernelGGL(k_verdict, dim3((unsigned)std::min<long long>(n_retry, h->grid_v)), dim3(64), h->lds_v, st, h->Pv,
                                   h->frontier.as<int32_t>(), (long long)n_retry, k, h->status.as<uint8_t>(), ctr, part_list(0), (const int32_t *)nullptr);
                HIP_TRY(h, hipGetLastError());
                if (region_launched) HIP_TRY(h, hipStreamWaitEvent(st, h->ev_rjoin, 0));
                if (r2_drained) HIP_TRY(h, hipStreamWaitEvent(st, h->ev_rjoin2, 0));
                { int rcs = recession(); if (rcs) return rcs; }
                { int rcs = partition({{ST_OPT_PENDING, 2}}, cntB); if (rcs) return rcs; }   // they may have turned out optimal
            }
            if (region_launched) {
                n_late = cntB[2];
                const int32_t *late = part_list(2);
                if (n_late > region_extra)
                    return fail(h, MPC_ERR_CAPACITY, "late optimal candidates (" + std::to_string(n_late) + ") exceed the spare region slots (" + std::to_string(region_extra) +
                                                     ") of the overlapped region stage: repeat the solve after mpc_set_region_overlap(h, 0)");
                if (n_late > 0) {
                    const int32_t n_a = (int32_t)h->n_opt;
                    HIP_TRY(h, hipMemcpyAsync(h->opt_list.as<int32_t>() + n_a, late, (size_t)n_late * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
                    hipLaunchKernelGGL(k_init_slots, dim3((unsigned)((n_late + 255) / 256)), dim3(256), 0, st, region_out_hi, h->fi, n_a, n_late, ST_RETRY, late);
                    HIP_TRY(h, hipGetLastError());
                    h->n_opt = n_a + n_late;
                }
                n_opt_fast = (int32_t)h->n_opt;
            } else {
                n_opt_fast = cntB[2];
                h->opt_ptr = part_list(2);   // stays valid until the next level's partition (the region fetch reads it before that)
            }
        } else {
            hipLaunchKernelGGL(k_verdict, dim3((unsigned)std::min<long long>(n, h->grid_v)), dim3(64), h->lds_v, st, h->Pv,
                               h->frontier.as<int32_t>(), n, k, h->status.as<uint8_t>(), ctr, (const int32_t *)nullptr, (const int32_t *)nullptr);
            HIP_TRY(h, hipGetLastError());
            { int rcs = recession(); if (rcs) return rcs; }
        }
        if (!tail_done) if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], st));
        // optimal candidates -> region kernel
        int32_t n_opt = n_opt_fast;
        if (n_opt < 0) {
            h->opt_ptr = h->opt_list.as<int32_t>();
            hipLaunchKernelGGL(k_flag_status, dim3(blocks256), dim3(256), 0, st, h->status.as<uint8_t>(), n, ST_OPT_PENDING, ST_OPT_PENDING, h->flag.as<int32_t>());
            { int rcs = launch_scan(h, h->flag.as<int32_t>(), h->pos.as<int32_t>(), n, total); if (rcs) return rcs; }
            hipLaunchKernelGGL(k_scatter_index, dim3(blocks256), dim3(256), 0, st, h->flag.as<int32_t>(), h->pos.as<int32_t>(), n, h->opt_list.as<int32_t>());
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(st));
            n_opt = h->tot_host[0];
        }
        h->n_opt = n_opt;
        if (tail_done) { /* the overlapped region stage left nothing to do: see the speculative end of the level above */ }
        else if (n_opt > 0 && h->fast && h->fast_r >= 0 && !h->force_v1) {
            if (!region_launched) { int rcs = region2_launch(n_opt, 0, st, false); if (rcs) return rcs; }
            // candidates k_region2 gave up on (counted by the kernel; normally none): the LDS-engine kernel, fixed-stride records
            hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, st, &ctr->n_rretry, reinterpret_cast<unsigned int *>(h->tot_dev + 8), 1);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(st));
            const unsigned int n_rr_dev = (unsigned int)h->tot_host[8];
            int32_t n_rr = 0;
            if (n_rr_dev > 0) { int rcs = compact(ST_RRETRY, ST_RRETRY, &n_rr); if (rcs) return rcs; }
            if (n_late > 0) {   // the late optimal candidates follow, in the order of their slots
                HIP_TRY(h, h->retry_list.ensure(nn * sizeof(int32_t), st));
                HIP_TRY(h, hipMemcpyAsync(h->retry_list.as<int32_t>() + n_rr, h->opt_list.as<int32_t>() + (n_opt - n_late), (size_t)n_late * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
                n_rr += n_late;
            }
            h->n_rretry = n_rr;
            if (n_rr > 0) {   // numerically doubtful regions: the LDS-engine kernel, fixed-stride records
                int rcs = launch_region_v1(h, h->retry_list.as<int32_t>(), n_rr, k, ctr);
                if (rcs) return rcs;
            }
        } else if (n_opt > 0) {
            int rcs = launch_region_v1(h, h->opt_ptr, n_opt, k, ctr);
            if (rcs) return rcs;
        }
        if (!tail_done) {
            if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], st));
            // pruned masks of this level + children, histogram, counters
            { int rcs = queue_tail(); if (rcs) return rcs; }
            HIP_TRY(h, hipStreamSynchronize(st));
        }
        if (prep_any || rprep_any) return fail(h, MPC_ERR_STATE, "level_run: a queued preparation (clear / copy) was never issued");
        h->r3_dirty = false;   // the main stream waited for ev_rjoin before the second partition
        std::memcpy(&host_ctr, h->tot_host + 16, sizeof(LevelCounters));
        h->n_r2_early = 0;
        if (r2_early_A) {
            if (host_ctr.q_fault) return fail(h, MPC_ERR_HIP, "region stage, queue form: a claimed queue entry never arrived");
            h->n_r2_early = host_ctr.q_early;
            // the last chunk of slots is a partial one the kernel could not recognise as complete (it never knew the queue's final length):
            // every slot is complete now
            if (h->so.active && h->st_flags.p) {
                int32_t *fl = h->st_flags.as<int32_t>();
                for (int j = 0; j < h->so.n_chunks; ++j) __atomic_store_n(fl + j, 1, __ATOMIC_RELEASE);
            }
            if (std::getenv("MPC_DEBUG_R2_EARLY")) std::fprintf(stderr, "[mpc] k=%d queue form of the region stage: %u of %lld regions' candidates built by the early launch\n", k, host_ctr.q_early, (long long)h->n_opt);
        }
        {
            const int32_t *cnt_host = h->tot_host + 16 + (int)(sizeof(LevelCounters) / 4);
            if (children_lean) h->n_children = cnt_host[20];
            if (theta_lean) n_theta_items = cnt_host[0];
            if (xq_lean) n_x_items = (long long)cnt_host[8] + (n_x_items - n_xq_items);   // what the quick test left + the dictionary-only items
        }
        if (h->timing) HIP_TRY(h, hipEventElapsedTime(&ms[0], h->ev[0], h->ev[1]));
        if (h->timing) HIP_TRY(h, hipEventElapsedTime(&ms[1], h->ev[1], h->ev[2]));
        if (h->timing) HIP_TRY(h, hipEventElapsedTime(&ms[2], h->ev[2], h->ev[3]));
        for (int i = 0; i < 5; ++i) if (kernel_timed[i] && h->timing) HIP_TRY(h, hipEventElapsedTime(&kms[i], h->kev[2 * i], h->kev[2 * i + 1]));
        if ((xq_thread_timed) && h->timing) { HIP_TRY(h, hipEventElapsedTime(&h->ms_xq_thread, h->kev[10], h->kev[11])); h->n_xq_thread = host_ctr.xq_thread; }
        h->ms_x1 = 0;
        if ((x1_ran) && h->timing) HIP_TRY(h, hipEventElapsedTime(&h->ms_x1, h->kev[12], h->kev[13]));
        h->n_x1 = x1_ran ? (long long)(h->tot_host + 16 + (int)(sizeof(LevelCounters) / 4))[12] : 0;   // dictionaries k_x1 wrote
        if (h->debug_cycles && x1_ran) { const int32_t *ch = h->tot_host + 16 + (int)(sizeof(LevelCounters) / 4); std::fprintf(stderr, "[mpc] k=%d one-step plans: %d streamed by k_x1, %d + %d + %d left to k_x2 (plan pass %.3f ms)\n", k, ch[12], ch[13], ch[14], ch[15], h->ms_xq_thread); }
        if (xq_early_ran) { n_xq_items += host_ctr.xq_thread; h->n_needx += host_ctr.xq_thread; }   // what the pass beside the theta stage decided never reached the partition's count
        if (kernel_timed[2] && host_ctr.r2_t1 > ~host_ctr.r2_not_t0 && h->wall_khz > 0)   // k_region2 times itself (see the kernel)
            kms[2] = (float)((double)(host_ctr.r2_t1 - ~host_ctr.r2_not_t0) / (double)h->wall_khz);
        if (h->debug_cycles)
            std::fprintf(stderr, "[mpc] k=%d n=%lld cycles/cand: kkt %.0f theta %.0f (rows %.0f, stage2 %.0f) x %.0f region %.0f; pivots %.2f; box screen %.3f / %.3f; x quick %.3f; retries theta %u of %llu\n", k, n,
                         host_ctr.cycles[0] / (double)n, host_ctr.cycles[1] / (double)n, host_ctr.cycles[4] / (double)n, host_ctr.cycles[5] / (double)n,
                         host_ctr.cycles[2] / (double)n, host_ctr.cycles[3] / (double)n, host_ctr.pivots / (double)n, host_ctr.cycles[6] / (double)n, host_ctr.cycles[7] / (double)n, host_ctr.xtheta_lps / (double)n, host_ctr.n_retry_theta, (unsigned long long)host_ctr.xtheta_fallbacks);
        if (h->debug_cycles && h->n_opt > 0)
            std::fprintf(stderr, "[mpc] k=%d region2 per optimal candidate (%lld): rows %.0f chebyshev %.0f facets %.0f total %.0f cycles; refactors %.2f facet pivots %.1f box-screened rows %.1f\n", k,
                         h->n_opt, host_ctr.rcycles[0] / (double)h->n_opt, host_ctr.rcycles[1] / (double)h->n_opt, host_ctr.rcycles[2] / (double)h->n_opt,
                         host_ctr.rcycles[3] / (double)h->n_opt, host_ctr.rcycles[4] / (double)h->n_opt, host_ctr.rcycles[5] / (double)h->n_opt, host_ctr.r_box / (double)h->n_opt);
        h->n_pruned_new = host_ctr.n_pruned_new;
        h->n_erows = host_ctr.e_rows;
        h->n_regions = (long long)host_ctr.status[ST_REGION];
    }
    graveyard_flush(h);   // (behind the level's closing synchronisation)
    h->level_done = true;
    h->last_level_n = n;
    stream_ready(h);
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->n = n; stats->k = k; stats->kkt_mode = h->kkt_mode;
        for (int i = 0; i < 6; ++i) stats->n_status[i] = (int64_t)host_ctr.status[i];
        stats->n_regions = h->n_regions; stats->n_children = h->n_children; stats->n_pruned_new = h->n_pruned_new;
        stats->lp_pivots = (int64_t)host_ctr.pivots;
        stats->n_xtheta_lp = (h->fast && !h->force_v1) ? h->n_needx : (int64_t)host_ctr.xtheta_lps;
        stats->n_xtheta_fallback = (int64_t)host_ctr.xtheta_fallbacks;
        for (int i = 0; i < 4; ++i) stats->wave_cycles[i] = (int64_t)host_ctr.cycles[i];
        stats->n_region_retry = h->n_rretry;
        stats->n_x_cached = (int64_t)host_ctr.x_cached;
        stats->ms_theta = kms[0]; stats->ms_x = kms[1]; stats->ms_region2 = kms[2];
        stats->region_side_stream = h->region_side_stream ? 1.0f : 0.0f;
        stats->n_x_items = n_x_items;
        stats->n_theta_items = n_theta_items;
        stats->ms_kkt = kms[3]; stats->ms_xq = kms[4];
        stats->n_xq_items = n_xq_items; stats->xq_pivots = (int64_t)host_ctr.xq_pivots;
        stats->xq_record_ints = dict_ints_head(h->Pf.n_d0r, h->fast_x >= 2 ? 32 : 16) - 1; stats->xq_record_rows = h->Pf.n_d0r; stats->xq_record_cols = h->Pf.n_d0c + 1;
        stats->n_xq_thread = h->n_xq_thread; stats->ms_xq_thread = h->ms_xq_thread; stats->xq_thread_beside_theta = xq_early_ran ? 1.0f : 0.0f;
        stats->n_x1 = h->n_x1; stats->ms_x1 = h->ms_x1; stats->ms_x_plan = x1_ran ? h->ms_xq_thread : 0.0f;
        stats->n_region_rows = h->n_erows;
        stats->n_opt = h->n_opt;
        // bytes of one dictionary record that are actually moved: the used columns (value + D0 columns) and the integer part
        const long long rec_bytes = (long long)(h->Pf.n_d0c + 1) * h->Pf.n_d0r * 8 + (long long)dict_ints_head(h->Pf.n_d0r, h->fast_x >= 2 ? 32 : 16) * 4;   // (what k_x2 reads of a record; the stored record also carries the children's look-up bytes)
        stats->dict_read_bytes = (h->fast && h->have_prev_dict && h->have_parent_slot) ? rec_bytes : 0;
        stats->dict_write_bytes = (h->fast && h->storing) ? rec_bytes : 0;
        stats->ms_verdict = ms[0]; stats->ms_region = ms[1]; stats->ms_children = ms[2]; stats->ms_total = ms[0] + ms[1] + ms[2];
    }
    return MPC_OK;
}

int mpc_level_run(mpc_handle *h, int32_t gen_children, mpc_level_stats *stats) { return mpc_level_run_ex(h, gen_children, 0, stats); }
int mpc_level_run_ex(mpc_handle *h, int32_t gen_children, int32_t flags, mpc_level_stats *stats) {
    if (!h) return MPC_ERR_INVALID;
    if ((flags & MPC_LEVEL_GRAPH) && gen_children) return fail(h, MPC_ERR_INVALID, "MPC_LEVEL_GRAPH has no children");
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "a level started with mpc_level_start is still running"); }
    // MPC_LEVEL_STREAM is honoured here too: the region kernel then writes its records into page-locked host memory and
    // mpc_level_stream_info hands the (complete) arrays over after this call -- no device-to-host fetch for small levels
    { std::lock_guard<std::mutex> lk(h->wm); h->w_stream_ready = false; h->a_ready.store(0, std::memory_order_release); }
    return level_run_impl(h, gen_children, flags & ~(MPC_LEVEL_THEN_BASE | MPC_LEVEL_ONLY_BASE), stats);
}

// ---- the same level, driven by the handle's worker thread ---------------------------------------------------------------
static int level_regions_slots_impl(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows, int64_t cap_rows,
                                    int64_t *n_slots, int64_t *n_rows, bool may_return_early, bool in_place);
// The base active set (the equality rows alone; driver :142-146) on the worker thread.  Any failure simply leaves the check to the caller.
static void worker_base_check(mpc_handle *h) {
    std::vector<int32_t> base((size_t)std::max(h->n_eq, 1));
    for (int i = 0; i < h->n_eq; ++i) base[(size_t)i] = i;
    mpc_level_stats bs;
    std::memset(&bs, 0, sizeof(bs));
    h->base_rec_d.assign((size_t)h->rec_d, 0.0);
    h->base_rec_i.assign((size_t)h->rec_i, -1);
    int64_t cand = 0;
    int rb = mpc_frontier_set(h, base.data(), 1, h->n_eq);
    if (rb == MPC_OK) { h->n_pruned = 0; rb = level_run_impl(h, 0, 0, &bs); }
    if (rb == MPC_OK) rb = mpc_level_status(h, &h->base_status);
    if (rb == MPC_OK && bs.n_regions > 0) rb = mpc_level_regions(h, h->base_rec_d.data(), h->base_rec_i.data(), &cand, 1);
    if (rb == MPC_OK) { h->base_regions = bs.n_regions; h->base_valid = true; }
}

// blocks of levels of the previous solve loop that nobody took
static void solve_release(mpc_handle *h) {
    if (!h->sv_levels) return;
    const int nl = h->sv_n.load(std::memory_order_acquire);
    // a level's k_fetch_slots is queued behind that level's closing synchronisation: a block nobody took may still be written to
    // (abandoned solve restarted at once) -- the stream is drained before such a block goes back to the pool
    bool pending = false;
    for (int i = 0; i < nl; ++i) {
        const auto &lv = h->sv_levels[i];
        if (!lv.handed && lv.ready_flag && !__atomic_load_n(lv.ready_flag, __ATOMIC_ACQUIRE)) pending = true;
    }
    if (pending && h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < nl; ++i) {
        auto &lv = h->sv_levels[i];
        if (!lv.handed) { if (lv.hd) (void)host_pool_give(lv.hd); if (lv.hi) (void)host_pool_give(lv.hi); if (lv.er) (void)host_pool_give(lv.er); }
        lv.hd = lv.hi = lv.er = nullptr; lv.handed = false; lv.flags = nullptr; lv.ready_flag = nullptr; lv.mode = 0;
        lv.ready.store(0, std::memory_order_relaxed); lv.done.store(0, std::memory_order_relaxed);
    }
    h->sv_n.store(0, std::memory_order_release);
}

// The level loop of the reference's driver (mp_solvers/mpqp_parrallel_combinatorial.py:98-139) on the worker thread: root frontier,
// then level after level -- run, publish the level's records, advance the frontier -- without a hand-over to the caller in between.
static int worker_solve(mpc_handle *h) {
    const int flags = h->sv_flags;
    int rc = mpc_pruned_clear(h);
    if (rc == MPC_OK) rc = mpc_frontier_root(h);
    const int level_flags = flags & (MPC_LEVEL_STREAM | MPC_LEVEL_KEEP_LOWDIM);
    for (int depth = 0; rc == MPC_OK && depth < h->sv_max_levels; ++depth) {
        const int gen = depth + 1 != h->sv_max_levels;
        auto &lv = h->sv_levels[depth];
        lv.k = h->k; lv.n = h->n;
        const auto t0 = std::chrono::steady_clock::now();
        h->sv_cur = depth;
        h->sv_n.store(depth + 1, std::memory_order_release);
        rc = level_run_impl(h, gen, level_flags, &lv.stats);
        if (rc == MPC_OK && !lv.ready.load(std::memory_order_relaxed)) {
            // the level did not stream (no optimal candidate; run without host round trips, its records in device buffers; LDS-engine
            // region kernel; > 1 GiB of records): fetched here, complete when published
            if ((flags & MPC_SOLVE_FETCH) && lv.stats.n_regions > 0 && h->n_opt > 0 && !h->so.active) {
                int64_t rows_cap = 0, ns = 0, nr = 0;
                mpc_compact_strides(h, nullptr, nullptr, &rows_cap);
                const size_t b_hd = (size_t)h->n_opt * h->fd * sizeof(double), b_hi = (size_t)h->n_opt * h->fi * sizeof(int32_t),
                             b_er = (size_t)std::max<int64_t>(rows_cap, 1) * (h->n_t + 1) * sizeof(double);
                const bool all_slots = h->used_region2 && h->n_rretry == 0 && !h->no_fetch_kernel;   // every record is in slot form on the device
                if (host_pool_take(b_hd, &lv.hd, nullptr, all_slots) != hipSuccess || host_pool_take(b_hi + 64, &lv.hi, nullptr, all_slots) != hipSuccess ||
                    host_pool_take(b_er, &lv.er, nullptr, all_slots) != hipSuccess) rc = fail(h, MPC_ERR_HIP, "mpc_solve: page-locked memory for a level's region records");
                if (rc == MPC_OK && all_slots) {
                    // one launch writes the three arrays into the host blocks and raises the flag behind head_i; nobody waits here
                    FetchCopy fc{};
                    int32_t *flag = reinterpret_cast<int32_t *>(static_cast<char *>(lv.hi) + ((b_hi + 7) & ~size_t(7)));
                    *flag = 0;
                    void *d_hd = nullptr, *d_hi = nullptr, *d_er = nullptr;
                    if (hipHostGetDevicePointer(&d_hd, lv.hd, 0) != hipSuccess || hipHostGetDevicePointer(&d_hi, lv.hi, 0) != hipSuccess ||
                        hipHostGetDevicePointer(&d_er, lv.er, 0) != hipSuccess) rc = fail(h, MPC_ERR_HIP, "hipHostGetDevicePointer (level records)");
                    if (rc == MPC_OK) {
                        fc.src[0] = h->headi.p; fc.dst[0] = d_hi; fc.bytes[0] = b_hi;
                        fc.src[1] = h->headd.p; fc.dst[1] = d_hd; fc.bytes[1] = b_hd;
                        fc.src[2] = h->epool.p; fc.dst[2] = d_er; fc.bytes[2] = (size_t)h->n_erows * (h->n_t + 1) * sizeof(double);
                        fc.counter = reinterpret_cast<unsigned int *>(h->dcnt.as<int32_t>() + 30);   // zero: cleared with the list lengths at the start of every level
                        fc.flag = reinterpret_cast<int32_t *>(static_cast<char *>(d_hi) + ((b_hi + 7) & ~size_t(7)));
                        const unsigned long long words = (b_hi + b_hd + fc.bytes[2]) / 4;
                        hipLaunchKernelGGL(k_fetch_slots, dim3((unsigned)std::min<unsigned long long>(256, words / 2048 + 1)), dim3(256), 0, h->stream, fc);
                        if (hipGetLastError() != hipSuccess) rc = fail(h, MPC_ERR_HIP, "k_fetch_slots launch");
                        lv.ready_flag = flag; ns = h->n_opt; nr = h->n_erows;
                    }
                } else if (rc == MPC_OK)
                    rc = level_regions_slots_impl(h, static_cast<double *>(lv.hd), static_cast<int32_t *>(lv.hi), h->n_opt, static_cast<double *>(lv.er),
                                                  rows_cap, &ns, &nr, false, false);
                lv.mode = 2; lv.n_slots = ns; lv.n_rows = nr; lv.chunk = 0; lv.n_chunks = 0;
            } else lv.mode = 0;
        } else if (rc == MPC_OK && lv.stats.n_region_retry > 0) {
            // candidates the LDS-engine kernel re-solved after the stream: their slots are filled in place (the caller lists the level
            // again after mpc_solve_level_wait)
            int64_t ns = 0, nr = 0;
            rc = level_regions_slots_impl(h, static_cast<double *>(lv.hd), static_cast<int32_t *>(lv.hi), lv.n_slots, static_cast<double *>(lv.er), lv.n_rows, &ns, &nr, false, true);
        }
        lv.ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        h->sv_cur = -1;
        if (rc != MPC_OK) break;
        lv.ready.store(1, std::memory_order_release);
        lv.done.store(1, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(h->wm); }
        h->wcv.notify_all();
        if (!gen || lv.stats.n_children == 0) break;
        rc = mpc_frontier_advance(h);
    }
    h->sv_cur = -1;
    h->base_valid = false;
    if (rc == MPC_OK && (flags & MPC_LEVEL_THEN_BASE)) worker_base_check(h);
    return rc;
}

static void worker_main(mpc_handle *h) {
    for (;;) {
        int gen = 0, flags = 0, req = 0;
        {
            for (int spin = 0; spin < 20000 && !h->a_req.load(std::memory_order_acquire); ++spin) __builtin_ia32_pause();
            std::unique_lock<std::mutex> lk(h->wm);
            h->wcv.wait(lk, [&] { return h->w_req != 0; });
            if (h->w_req == 2) return;
            req = h->w_req; gen = h->w_gen; flags = h->w_flags; h->w_req = 0; h->a_req.store(0, std::memory_order_release);
        }
        mpc_level_stats st;
        std::memset(&st, 0, sizeof(st));
        if (req == 3) {
            const int rcs = worker_solve(h);
            {
                std::lock_guard<std::mutex> lk(h->wm);
                h->sv_rc = rcs; h->w_rc = rcs; h->w_stats = st; h->w_busy = false; h->w_stream_ready = true;
                h->sv_finished.store(1, std::memory_order_release);
                h->a_ready.store(1, std::memory_order_release); h->a_busy.store(0, std::memory_order_release);
            }
            h->wcv.notify_all();
            continue;
        }
        const bool only_base = (flags & MPC_LEVEL_ONLY_BASE) != 0;
        const int rc = only_base ? MPC_OK : level_run_impl(h, gen, flags, &st);
        h->base_valid = false;
        // only when the level's records were streamed AND the caller already owns the arrays (or the level has no region at all): a
        // level that did not stream (LDS-engine region kernel ...) is fetched by the caller after mpc_level_wait, from the very state
        // this check would replace
        const bool streamed_and_taken = h->so.active && h->so.taken;
        if (rc == MPC_OK && (only_base || ((flags & MPC_LEVEL_THEN_BASE) && !gen && st.n_region_retry == 0 && (streamed_and_taken || st.n_regions == 0))))
            worker_base_check(h);   // right behind the last level, while the caller is still turning the streamed records into objects
        {
            std::lock_guard<std::mutex> lk(h->wm);
            h->w_rc = rc; h->w_stats = st; h->w_busy = false; h->w_stream_ready = true;
            h->a_ready.store(1, std::memory_order_release); h->a_busy.store(0, std::memory_order_release);
        }
        h->wcv.notify_all();
    }
}

int mpc_level_start(mpc_handle *h, int32_t gen_children, int32_t flags) {
    if (!h) return MPC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(h->wm);
    if (h->w_busy) return fail(h, MPC_ERR_STATE, "mpc_level_start: the previous level has not been waited for");
    if (!h->worker.joinable()) h->worker = std::thread(worker_main, h);
    h->w_gen = gen_children; h->w_flags = flags; h->w_busy = true; h->w_stream_ready = false; h->w_req = 1;
    h->a_busy.store(1, std::memory_order_release); h->a_ready.store(0, std::memory_order_release); h->a_req.store(1, std::memory_order_release);
    lk.unlock();
    h->wcv.notify_all();
    return MPC_OK;
}

int mpc_level_stream_info(mpc_handle *h, double **head_d, int32_t **head_i, double **erows, int64_t *n_slots, int64_t *cap_rows,
                          int32_t *chunk, int32_t *n_chunks) {
    if (!h) return MPC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(h->wm);
    if (!h->w_busy && !h->w_stream_ready) return fail(h, MPC_ERR_STATE, "mpc_level_stream_info without a level started by mpc_level_start");
    if (!h->w_stream_ready) {
        lk.unlock();
        for (int spin = 0; spin < 200000 && !h->a_ready.load(std::memory_order_acquire); ++spin) __builtin_ia32_pause();
        lk.lock();
    }
    h->wcv.wait(lk, [&] { return h->w_stream_ready; });
    auto &so = h->so;
    const bool on = so.active && !so.taken;
    if (head_d) *head_d = on ? static_cast<double *>(so.hd) : nullptr;
    if (head_i) *head_i = on ? static_cast<int32_t *>(so.hi) : nullptr;
    if (erows) *erows = on ? static_cast<double *>(so.er) : nullptr;
    if (n_slots) *n_slots = on ? so.n_slots : 0;
    if (cap_rows) *cap_rows = on ? so.cap_rows : 0;
    if (chunk) *chunk = on ? (1 << so.shift) : 0;
    if (n_chunks) *n_chunks = on ? so.n_chunks : 0;
    if (on) so.taken = true;   // the three blocks now belong to the caller (mpc_host_free)
    return MPC_OK;
}

int mpc_level_chunk_wait(mpc_handle *h, int32_t j) {
    if (!h || j < 0 || j >= h->cw_chunks || !h->st_flags.p) return MPC_ERR_INVALID;
    const int32_t *flag = h->st_flags.as<int32_t>() + j;
    for (unsigned spin = 0;; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE)) return MPC_OK;
        if ((spin & 1023u) == 1023u) {
            // the level has ended (normally or with an error) without raising the flag: do not wait for ever
            bool busy;
            { std::lock_guard<std::mutex> lk(h->wm); busy = h->w_busy; }
            if (!busy) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) ? MPC_OK : fail(h, MPC_ERR_STATE, "mpc_level_chunk_wait: the level ended without completing this chunk");
        }
        __builtin_ia32_pause();
    }
}

int mpc_level_wait(mpc_handle *h, mpc_level_stats *stats) {
    if (!h) return MPC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(h->wm);
    if (!h->worker.joinable()) return fail(h, MPC_ERR_STATE, "mpc_level_wait without mpc_level_start");
    if (h->w_busy) {
        lk.unlock();
        for (int spin = 0; spin < 200000 && h->a_busy.load(std::memory_order_acquire); ++spin) __builtin_ia32_pause();
        lk.lock();
    }
    h->wcv.wait(lk, [&] { return !h->w_busy; });
    if (stats) *stats = h->w_stats;
    return h->w_rc;
}

// ---- the whole level loop behind one call ---------------------------------------------------------------------------------------
int mpc_solve_start(mpc_handle *h, int32_t max_levels, int32_t flags) {
    if (!h || max_levels < 0) return MPC_ERR_INVALID;
    std::unique_lock<std::mutex> lk(h->wm);
    if (h->w_busy) return fail(h, MPC_ERR_STATE, "mpc_solve_start: the previous level / solve has not been waited for");
    if (!h->sv_levels) h->sv_levels.reset(new mpc_handle::SolveLevel[MPC_MAX_NC + 2]);
    solve_release(h);
    h->sv_max_levels = std::min<int>(max_levels, MPC_MAX_NC + 1); h->sv_flags = flags; h->sv_rc = MPC_OK;
    h->sv_finished.store(0, std::memory_order_release);
    if (!h->worker.joinable()) h->worker = std::thread(worker_main, h);
    h->w_gen = 0; h->w_flags = flags; h->w_busy = true; h->w_stream_ready = false; h->w_req = 3;
    h->a_busy.store(1, std::memory_order_release); h->a_ready.store(0, std::memory_order_release); h->a_req.store(1, std::memory_order_release);
    lk.unlock();
    h->wcv.notify_all();
    return MPC_OK;
}

// waits (short spin, then the condition variable) until pred() holds
static void solve_wait_for(mpc_handle *h, const std::function<bool()> &pred) {
    for (int spin = 0; spin < 400000; ++spin) { if (pred()) return; __builtin_ia32_pause(); }
    std::unique_lock<std::mutex> lk(h->wm);
    h->wcv.wait(lk, pred);
}

int mpc_solve_level(mpc_handle *h, int32_t level, mpc_solve_level_info *info) {
    if (!h || !info || level < 0 || !h->sv_levels || level > MPC_MAX_NC) return MPC_ERR_INVALID;
    auto &lv = h->sv_levels[level];
    solve_wait_for(h, [&] { return lv.ready.load(std::memory_order_acquire) != 0 || h->sv_finished.load(std::memory_order_acquire) != 0; });
    std::memset(info, 0, sizeof(*info));
    info->level = level;
    if (!lv.ready.load(std::memory_order_acquire)) { info->mode = -1; return h->sv_rc; }   // the loop ended before (or inside) this level
    if (lv.mode == 2 && lv.ready_flag) {
        // the arrays are being written by k_fetch_slots on the handle's stream: its last workgroup raises the flag
        for (unsigned spin = 0; !__atomic_load_n(lv.ready_flag, __ATOMIC_ACQUIRE); ++spin) {
            if ((spin & 4095u) == 4095u && h->sv_finished.load(std::memory_order_acquire)) {
                if (h->sv_rc != MPC_OK) { info->mode = -1; return h->sv_rc; }
                // the loop has ended without an error: the copy kernel is queued on the handle's stream -- once that stream has drained the
                // flag is either up or will never be (a launch that failed later than hipGetLastError could see)
                (void)hipSetDevice(h->device);
                const hipError_t es = hipStreamSynchronize(h->stream);
                if (es != hipSuccess || !__atomic_load_n(lv.ready_flag, __ATOMIC_ACQUIRE)) {
                    info->mode = -1;
                    return fail(h, es != hipSuccess ? MPC_ERR_HIP : MPC_ERR_STATE, "mpc_solve_level: the solve ended without completing this level's record copy");
                }
            }
            __builtin_ia32_pause();
        }
    }
    info->k = lv.k; info->n = lv.n; info->mode = lv.mode; info->chunk = lv.chunk; info->n_chunks = lv.n_chunks;
    info->n_slots = lv.n_slots; info->n_rows = lv.n_rows;
    if (lv.mode != 0 && !lv.handed) {
        info->head_d = static_cast<double *>(lv.hd); info->head_i = static_cast<int32_t *>(lv.hi); info->erows = static_cast<double *>(lv.er);
        lv.handed = true;   // the three blocks now belong to the caller (mpc_host_free)
    } else if (lv.mode != 0) return fail(h, MPC_ERR_STATE, "mpc_solve_level: the arrays of this level have been handed over already");
    return MPC_OK;
}

int mpc_solve_chunk_wait(mpc_handle *h, int32_t level, int32_t j) {
    if (!h || level < 0 || !h->sv_levels || level > MPC_MAX_NC) return MPC_ERR_INVALID;
    auto &lv = h->sv_levels[level];
    if (lv.mode != 1 || j < 0 || j >= lv.n_chunks) return MPC_ERR_INVALID;
    for (unsigned spin = 0;; ++spin) {
        // a finished level has every chunk complete (and its flag words may already belong to the next level)
        if (lv.done.load(std::memory_order_acquire)) return MPC_OK;
        if (__atomic_load_n(lv.flags + j, __ATOMIC_ACQUIRE)) return MPC_OK;
        if ((spin & 1023u) == 1023u && h->sv_finished.load(std::memory_order_acquire))
            return lv.done.load(std::memory_order_acquire) ? MPC_OK : fail(h, MPC_ERR_STATE, "mpc_solve_chunk_wait: the solve ended without completing this chunk");
        __builtin_ia32_pause();
    }
}

int mpc_solve_level_wait(mpc_handle *h, int32_t level, mpc_level_stats *stats, double *ms_wall) {
    if (!h || level < 0 || !h->sv_levels || level > MPC_MAX_NC) return MPC_ERR_INVALID;
    auto &lv = h->sv_levels[level];
    solve_wait_for(h, [&] { return lv.done.load(std::memory_order_acquire) != 0 || h->sv_finished.load(std::memory_order_acquire) != 0; });
    if (!lv.done.load(std::memory_order_acquire)) return h->sv_rc != MPC_OK ? h->sv_rc : fail(h, MPC_ERR_STATE, "mpc_solve_level_wait: the solve ended before this level");
    if (stats) *stats = lv.stats;
    if (ms_wall) *ms_wall = lv.ms_wall;
    return MPC_OK;
}

int mpc_solve_wait(mpc_handle *h, int32_t *n_levels) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->worker.joinable() || !h->sv_levels) return fail(h, MPC_ERR_STATE, "mpc_solve_wait without mpc_solve_start");
    {
        std::unique_lock<std::mutex> lk(h->wm);
        if (h->w_busy) {
            lk.unlock();
            for (int spin = 0; spin < 200000 && h->a_busy.load(std::memory_order_acquire); ++spin) __builtin_ia32_pause();
            lk.lock();
        }
        h->wcv.wait(lk, [&] { return !h->w_busy; });
    }
    if (n_levels) {
        int nl = 0;
        while (nl < h->sv_n.load(std::memory_order_acquire) && h->sv_levels[nl].done.load(std::memory_order_acquire)) ++nl;
        *n_levels = nl;
    }
    return h->sv_rc;
}

int mpc_base_result(mpc_handle *h, uint8_t *status, int64_t *n_regions, double *rec_d, int32_t *rec_i) {
    if (!h) return MPC_ERR_INVALID;
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "the level is still running"); }
    if (!h->base_valid) return fail(h, MPC_ERR_STATE, "no base-set result (mpc_level_start without MPC_LEVEL_THEN_BASE, or the check was left to the caller)");
    if (status) *status = h->base_status;
    if (n_regions) *n_regions = h->base_regions;
    if (h->base_regions > 0) {
        if (rec_d) std::memcpy(rec_d, h->base_rec_d.data(), sizeof(double) * h->base_rec_d.size());
        if (rec_i) std::memcpy(rec_i, h->base_rec_i.data(), sizeof(int32_t) * h->base_rec_i.size());
    }
    h->base_valid = false;
    return MPC_OK;
}

int mpc_level_status(mpc_handle *h, uint8_t *status) {
    if (!h || !status) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (h->n > 0) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(status, h->status.p, (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return MPC_OK;
}

int mpc_compact_strides(const mpc_handle *h, int64_t *fd, int64_t *fi, int64_t *max_rows) {
    if (!h) return MPC_ERR_INVALID;
    if (fd) *fd = h->n_x * h->n_t + h->n_x + h->k * h->n_t + h->k;
    if (fi) *fi = 8 + h->k + h->n_tc + h->k + 2 * (h->n_c - h->k);
    if (max_rows) {
        const long long rows_t = h->n_c - h->n_eq + h->n_tc;
        *max_rows = h->used_region2 ? h->n_erows + h->n_rretry * rows_t : h->n_regions * rows_t;
    }
    return MPC_OK;
}

// Regions of the level in compact form, frontier order, written straight into the caller's arrays.
int mpc_level_regions_compact(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_regions, double *erows, int64_t cap_rows,
                              int64_t *n_regions, int64_t *n_rows) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (n_regions) *n_regions = h->n_regions;
    if (n_rows) *n_rows = 0;
    if (h->so.active) return fail(h, MPC_ERR_STATE, "the records of this level were streamed to the host (mpc_level_stream_info)");
    if (cap_regions < h->n_regions) return fail(h, MPC_ERR_CAPACITY, "region buffers too small");
    if (h->n_regions == 0 || h->n_opt == 0) return MPC_OK;
    if (!head_d || !head_i || !erows) return MPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int nx = h->n_x, nt = h->n_t, nc = h->n_c, ntc = h->n_tc, k = h->k, nr = nt + 1, fd = h->fd, fi = h->fi;
    const long long n_opt = h->n_opt, rows_t = nc - h->n_eq + ntc;
    const long long n_fixed = h->used_region2 ? h->n_rretry : n_opt;
    hipStream_t s = h->stream;
    HIP_TRY(h, h->st_list.ensure((size_t)n_opt * sizeof(int32_t)));
    HIP_TRY(h, h->st_status.ensure((size_t)h->n));
    HIP_TRY(h, hipMemcpyAsync(h->st_list.p, h->opt_ptr, (size_t)n_opt * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(h->st_status.p, h->status.p, (size_t)h->n, hipMemcpyDeviceToHost, s));
    if (n_fixed > 0) {
        HIP_TRY(h, h->st_fxd.ensure((size_t)n_fixed * h->rec_d * sizeof(double)));
        HIP_TRY(h, h->st_fxi.ensure((size_t)n_fixed * h->rec_i * sizeof(int32_t)));
        HIP_TRY(h, hipMemcpyAsync(h->st_fxd.p, h->recd.p, (size_t)n_fixed * h->rec_d * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->st_fxi.p, h->reci.p, (size_t)n_fixed * h->rec_i * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    if (h->used_region2) {
        HIP_TRY(h, h->st_hd.ensure((size_t)n_opt * fd * sizeof(double)));
        HIP_TRY(h, h->st_hi.ensure((size_t)n_opt * fi * sizeof(int32_t)));
        HIP_TRY(h, h->st_pool.ensure((size_t)std::max<long long>(h->n_erows, 1) * nr * sizeof(double)));
        HIP_TRY(h, hipMemcpyAsync(h->st_hd.p, h->headd.p, (size_t)n_opt * fd * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->st_hi.p, h->headi.p, (size_t)n_opt * fi * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        if (h->n_erows > 0) HIP_TRY(h, hipMemcpyAsync(h->st_pool.p, h->epool.p, (size_t)h->n_erows * nr * sizeof(double), hipMemcpyDeviceToHost, s));
        if (h->n_rretry > 0) {
            HIP_TRY(h, h->st_rlist.ensure((size_t)h->n_rretry * sizeof(int32_t)));
            HIP_TRY(h, hipMemcpyAsync(h->st_rlist.p, h->retry_list.p, (size_t)h->n_rretry * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    const int32_t *list = h->st_list.as<int32_t>(), *rlist = h->st_rlist.as<int32_t>();
    const uint8_t *st = h->st_status.as<uint8_t>();
    const double *fxd = h->st_fxd.as<double>(), *hd = h->st_hd.as<double>(), *pool = h->st_pool.as<double>();
    const int32_t *fxi = h->st_fxi.as<int32_t>(), *hi = h->st_hi.as<int32_t>();
    long long wreg = 0, wrow = 0;
    bool overflow = false;
    // fixed record (mpcombi.h layout) -> compact head + rows
    auto from_fixed = [&](const double *rd, const int32_t *ri, int cand) {
        const int kk = ri[0], nE = ri[1], n_om = ri[2], n_la = ri[3], n_re = ri[4];
        if (wrow + nE > cap_rows) { overflow = true; return; }
        double *od = head_d + (size_t)wreg * fd;
        int32_t *oi = head_i + (size_t)wreg * fi;
        std::fill(od, od + fd, 0.0); std::fill(oi, oi + fi, -1);
        std::memcpy(od, rd, sizeof(double) * (nx * nt + nx));
        const double *Al = rd + nx * nt + nx, *bl = Al + (size_t)nc * nt, *E = bl + nc, *f = E + (size_t)(nc + ntc) * nt;
        std::memcpy(od + nx * nt + nx, Al, sizeof(double) * kk * nt);
        std::memcpy(od + nx * nt + nx + k * nt, bl, sizeof(double) * kk);
        oi[0] = ST_REGION; oi[1] = cand; oi[2] = nE; oi[3] = n_om; oi[4] = n_la; oi[5] = n_re; oi[6] = (int32_t)wrow; oi[7] = 0;
        int32_t *act = oi + 8, *om = act + k, *la = om + ntc, *ridx = la + k, *rcon = ridx + (nc - k);
        std::memcpy(act, ri + 5, sizeof(int32_t) * kk);
        std::memcpy(om, ri + 5 + nc, sizeof(int32_t) * n_om);
        std::memcpy(la, ri + 5 + nc + ntc, sizeof(int32_t) * n_la);
        std::memcpy(ridx, ri + 5 + nc + ntc + nc, sizeof(int32_t) * n_re);
        std::memcpy(rcon, ri + 5 + nc + ntc + nc + nc, sizeof(int32_t) * n_re);
        for (int r = 0; r < nE; ++r) {
            double *row = erows + (size_t)(wrow + r) * nr;
            row[0] = f[r];
            for (int t = 0; t < nt; ++t) row[1 + t] = E[(size_t)r * nt + t];
        }
        wrow += nE; ++wreg;
    };
    long long rpos = 0;
    for (long long w = 0; w < n_opt && !overflow && wreg < cap_regions; ++w) {
        const int cand = list[w];
        if (!h->used_region2) {
            if (st[cand] == ST_REGION) from_fixed(fxd + (size_t)w * h->rec_d, fxi + (size_t)w * h->rec_i, cand);
            continue;
        }
        const int32_t *src_i = hi + (size_t)w * fi;
        if (src_i[0] == ST_REGION) {
            const int nE = src_i[2], off = src_i[6];
            if (wrow + nE > cap_rows) { overflow = true; break; }
            std::memcpy(head_d + (size_t)wreg * fd, hd + (size_t)w * fd, sizeof(double) * fd);
            int32_t *oi = head_i + (size_t)wreg * fi;
            std::memcpy(oi, src_i, sizeof(int32_t) * fi);
            oi[6] = (int32_t)wrow;
            std::memcpy(erows + (size_t)wrow * nr, pool + (size_t)off * nr, sizeof(double) * nE * nr);
            wrow += nE; ++wreg;
        } else if (src_i[0] == ST_RETRY) {
            while (rpos < h->n_rretry && rlist[rpos] != cand) ++rpos;
            if (rpos < h->n_rretry && st[cand] == ST_REGION) from_fixed(fxd + (size_t)rpos * h->rec_d, fxi + (size_t)rpos * h->rec_i, cand);
        }
    }
    (void)rows_t;
    if (overflow) return fail(h, MPC_ERR_CAPACITY, "row buffer too small (see mpc_compact_strides max_rows)");
    if (n_regions) *n_regions = wreg;
    if (n_rows) *n_rows = wrow;
    return MPC_OK;
}

int64_t mpc_level_slots(const mpc_handle *h) { return h ? h->n_opt : 0; }

// in_place: the level's records were streamed into head_d / head_i / erows (host memory) by the region kernel itself; only
// the slots of candidates re-solved by the LDS-engine kernel still have to be filled in
static int level_regions_slots_impl(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows, int64_t cap_rows,
                                    int64_t *n_slots, int64_t *n_rows, bool may_return_early, bool in_place) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (h->so.active && !in_place) return fail(h, MPC_ERR_STATE, "the records of this level were streamed to the host (mpc_level_stream_info)");
    if (n_slots) *n_slots = 0;
    if (n_rows) *n_rows = 0;
    if (h->n_regions == 0 || h->n_opt == 0) return MPC_OK;
    if (cap_slots < h->n_opt) return fail(h, MPC_ERR_CAPACITY, "slot buffers too small (mpc_level_slots)");
    if (!head_d || !head_i || !erows) return MPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int nx = h->n_x, nt = h->n_t, nc = h->n_c, ntc = h->n_tc, k = h->k, nr = nt + 1, fd = h->fd, fi = h->fi;
    const long long n_opt = h->n_opt, rows_t = nc - h->n_eq + ntc;
    const bool merged = h->used_region2 && h->rretry_rows >= 0;   // k_rretry_merge has written the re-solved candidates' records into their slots
    const long long pool_rows = merged ? h->rretry_rows : h->n_erows;
    const long long n_fixed = h->used_region2 ? (merged ? 0 : h->n_rretry) : n_opt;
    const long long rows_need = h->used_region2 ? (merged ? pool_rows : h->n_erows + h->n_rretry * rows_t) : h->n_regions * rows_t;
    if (cap_rows < rows_need) return fail(h, MPC_ERR_CAPACITY, "row buffer too small (see mpc_compact_strides max_rows)");
    hipStream_t s = h->stream;
    if (n_fixed > 0) {
        HIP_TRY(h, h->st_fxd.ensure((size_t)n_fixed * h->rec_d * sizeof(double)));
        HIP_TRY(h, h->st_fxi.ensure((size_t)n_fixed * h->rec_i * sizeof(int32_t)));
        HIP_TRY(h, h->st_list.ensure((size_t)n_opt * sizeof(int32_t)));
        HIP_TRY(h, h->st_status.ensure((size_t)h->n));
        HIP_TRY(h, hipMemcpyAsync(h->st_fxd.p, h->recd.p, (size_t)n_fixed * h->rec_d * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->st_fxi.p, h->reci.p, (size_t)n_fixed * h->rec_i * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->st_list.p, h->opt_ptr, (size_t)n_opt * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipMemcpyAsync(h->st_status.p, h->status.p, (size_t)h->n, hipMemcpyDeviceToHost, s));
    }
    long long wrow = 0;
    if (h->used_region2 && n_fixed == 0 && may_return_early && !in_place) {
        // every record is in slot form on the device: the small integer heads first, then the two large arrays; the call
        // returns when the heads have arrived, the rest is in flight on the handle's stream (mpc_sync completes it)
        HIP_TRY(h, hipMemcpyAsync(head_i, h->headi.p, (size_t)n_opt * fi * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipEventRecord(h->ev_hi, s));
        HIP_TRY(h, hipMemcpyAsync(head_d, h->headd.p, (size_t)n_opt * fd * sizeof(double), hipMemcpyDeviceToHost, s));
        if (pool_rows > 0) HIP_TRY(h, hipMemcpyAsync(erows, h->epool.p, (size_t)pool_rows * nr * sizeof(double), hipMemcpyDeviceToHost, s));
        if (!h->fetch_nowait) HIP_TRY(h, hipEventSynchronize(h->ev_hi));
        if (n_slots) *n_slots = n_opt;
        if (n_rows) *n_rows = pool_rows;
        return MPC_OK;
    }
    if (h->used_region2) {
        if (!in_place) {
            HIP_TRY(h, hipMemcpyAsync(head_d, h->headd.p, (size_t)n_opt * fd * sizeof(double), hipMemcpyDeviceToHost, s));
            HIP_TRY(h, hipMemcpyAsync(head_i, h->headi.p, (size_t)n_opt * fi * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            if (pool_rows > 0) HIP_TRY(h, hipMemcpyAsync(erows, h->epool.p, (size_t)pool_rows * nr * sizeof(double), hipMemcpyDeviceToHost, s));
        }
        wrow = pool_rows;
        if (h->n_rretry > 0 && !merged) {
            HIP_TRY(h, h->st_rlist.ensure((size_t)h->n_rretry * sizeof(int32_t)));
            HIP_TRY(h, hipMemcpyAsync(h->st_rlist.p, h->retry_list.p, (size_t)h->n_rretry * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        }
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    if (h->debug_cycles && h->used_region2) {
        long long hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long long w = 0; w < n_opt; ++w) { const int32_t *oi = head_i + (size_t)w * fi; if (oi[0] == ST_RETRY) hist[oi[7] & 7]++; }
        std::fprintf(stderr, "[mpc] k=%d region retries by reason: chebyshev %lld, r->0 %lld, refactor-after-chebyshev %lld, row lost %lld, >8 refactors %lld, refactor failed %lld\n",
                     k, hist[1], hist[2], hist[3], hist[4], hist[5], hist[6]);
    }
    if (n_fixed > 0) {
        // the candidates solved by the LDS-engine kernel come as fixed-stride records: written into their slots here
        const int32_t *list = h->st_list.as<int32_t>(), *rlist = h->st_rlist.as<int32_t>();
        const uint8_t *st = h->st_status.as<uint8_t>();
        const double *fxd = h->st_fxd.as<double>();
        const int32_t *fxi = h->st_fxi.as<int32_t>();
        long long rpos = 0;
        for (long long w = 0; w < n_opt; ++w) {
            const int cand = list[w];
            int32_t *oi = head_i + (size_t)w * fi;
            long long src = -1;
            if (!h->used_region2) src = w;
            else if (oi[0] == ST_RETRY) {
                while (rpos < h->n_rretry && rlist[rpos] != cand) ++rpos;
                if (rpos < h->n_rretry) src = rpos;
            } else continue;
            std::fill(oi, oi + fi, -1);
            oi[0] = st[cand]; oi[1] = cand; oi[2] = oi[3] = oi[4] = oi[5] = oi[6] = oi[7] = 0;
            if (src < 0 || st[cand] != ST_REGION) continue;
            const double *rd = fxd + (size_t)src * h->rec_d;
            const int32_t *ri = fxi + (size_t)src * h->rec_i;
            const int kk = ri[0], nE = ri[1], n_om = ri[2], n_la = ri[3], n_re = ri[4];
            double *od = head_d + (size_t)w * fd;
            std::fill(od, od + fd, 0.0);
            std::memcpy(od, rd, sizeof(double) * (nx * nt + nx));
            const double *Al = rd + nx * nt + nx, *bl = Al + (size_t)nc * nt, *E = bl + nc, *f = E + (size_t)(nc + ntc) * nt;
            std::memcpy(od + nx * nt + nx, Al, sizeof(double) * kk * nt);
            std::memcpy(od + nx * nt + nx + k * nt, bl, sizeof(double) * kk);
            oi[2] = nE; oi[3] = n_om; oi[4] = n_la; oi[5] = n_re; oi[6] = (int32_t)wrow;
            int32_t *act = oi + 8, *om = act + k, *la = om + ntc, *ridx = la + k, *rcon = ridx + (nc - k);
            std::memcpy(act, ri + 5, sizeof(int32_t) * kk);
            std::memcpy(om, ri + 5 + nc, sizeof(int32_t) * n_om);
            std::memcpy(la, ri + 5 + nc + ntc, sizeof(int32_t) * n_la);
            std::memcpy(ridx, ri + 5 + nc + ntc + nc, sizeof(int32_t) * n_re);
            std::memcpy(rcon, ri + 5 + nc + ntc + nc + nc, sizeof(int32_t) * n_re);
            for (int r = 0; r < nE; ++r) {
                double *row = erows + (size_t)(wrow + r) * nr;
                row[0] = f[r];
                for (int t = 0; t < nt; ++t) row[1 + t] = E[(size_t)r * nt + t];
            }
            wrow += nE;
        }
    }
    if (n_slots) *n_slots = n_opt;
    if (n_rows) *n_rows = wrow;
    return MPC_OK;
}

int mpc_level_regions_slots(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows, int64_t cap_rows,
                            int64_t *n_slots, int64_t *n_rows) {
    return level_regions_slots_impl(h, head_d, head_i, cap_slots, erows, cap_rows, n_slots, n_rows, false, false);
}
int mpc_level_regions_slots_async(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows, int64_t cap_rows,
                                  int64_t *n_slots, int64_t *n_rows) {
    return level_regions_slots_impl(h, head_d, head_i, cap_slots, erows, cap_rows, n_slots, n_rows, true, false);
}
int mpc_level_regions_slots_nowait(mpc_handle *h, double *head_d, int32_t *head_i, int64_t cap_slots, double *erows, int64_t cap_rows,
                                   int64_t *n_slots, int64_t *n_rows) {
    if (!h) return MPC_ERR_INVALID;
    h->fetch_nowait = true;
    const int rc = level_regions_slots_impl(h, head_d, head_i, cap_slots, erows, cap_rows, n_slots, n_rows, true, false);
    h->fetch_nowait = false;
    return rc;
}
// one event per device: "the last k_fetch_many has finished" (recorded on the stream it ran on)
static std::mutex g_fetch_mutex;
static std::map<int, hipEvent_t> g_fetch_event;
// the launch's table, one pair of buffers PER DEVICE: the wait that frees a table for reuse is the same device's previous launch
static std::map<int, DevBuf> g_fetch_tab_dev_of;
static std::map<int, HostBuf> g_fetch_tab_host_of;
static int fetch_many_wait(int device) {
    // Every copy launch waits for the one before it (its table is reused), so the LATEST record of the device's event covers all earlier
    // launches: whoever waits -- from any thread, for any member -- waits for that record (a completed event returns at once).
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_fetch_mutex);
        auto it = g_fetch_event.find(device);
        if (it == g_fetch_event.end() || !it->second) return MPC_OK;
        ev = it->second;
    }
    return hipEventSynchronize(ev) == hipSuccess ? MPC_OK : MPC_ERR_HIP;
}
int mpc_fetch_wait(int32_t device) { return fetch_many_wait(device); }

int mpc_level_batch_fetch(mpc_handle **hs, int32_t n_handles, double *const *head_d, int32_t *const *head_i, const int64_t *cap_slots,
                          double *const *erows, const int64_t *cap_rows, int64_t *n_slots, int64_t *n_rows) {
    if (!hs || n_handles < 0 || !head_d || !head_i || !cap_slots || !erows || !cap_rows || !n_slots || !n_rows) return MPC_ERR_INVALID;
    static const bool no_many = [] { const char *ev = std::getenv("MPC_NO_FETCH_MANY"); return ev && ev[0] == '1'; }();   // A/B: three copy commands per member
    static const bool no_merge = [] { const char *ev = std::getenv("MPC_NO_RRETRY_MERGE"); return ev && ev[0] == '1'; }();   // A/B, tests: re-solved candidates' records merged by the host (round 4)
    auto no_merge_or = [&](bool plain) { return no_merge && !plain; };
    std::vector<FetchEntry> tab;
    mpc_handle *lead = nullptr;
    for (int i = 0; i < n_handles; ++i) {
        mpc_handle *h = hs[i];
        if (!h) return MPC_ERR_INVALID;
        // every record in slot form on the device, nothing streamed, nothing re-solved by the LDS engine: the copy kernel takes the member
        // (round 5: also a member with candidates re-solved by the LDS-engine kernel -- k_rretry_merge writes their records into the slots first)
        const long long rows_t = h->n_c - h->n_eq + h->n_tc;
        const bool merge = h->n_rretry > 0 && h->rretry_rows < 0;
        const long long rows_out = h->n_rretry > 0 ? h->n_erows + h->n_rretry * rows_t : h->n_erows;   // (bound: a re-solved region keeps at most rows_t rows)
        const bool fast = !no_many && !no_merge_or(h->n_rretry == 0) && h->level_done && !h->so.active && h->n_regions > 0 && h->n_opt > 0 && h->used_region2 &&
                          h->n_rretry <= RRETRY_MERGE_MAX && h->n_opt <= 0x7fffffffLL &&
                          (h->n_rretry == 0 || (h->recd.p && h->reci.p && h->retry_list.p && h->epool.cap >= (size_t)rows_out * (h->n_t + 1) * sizeof(double))) &&
                          cap_slots[i] >= h->n_opt && cap_rows[i] >= rows_out && head_d[i] && head_i[i] && (erows[i] || rows_out == 0) &&
                          (!lead || lead->device == h->device);
        void *d_hd = nullptr, *d_hi = nullptr, *d_er = nullptr;
        if (fast && hipHostGetDevicePointer(&d_hd, head_d[i], 0) == hipSuccess && hipHostGetDevicePointer(&d_hi, head_i[i], 0) == hipSuccess &&
            (rows_out == 0 || hipHostGetDevicePointer(&d_er, erows[i], 0) == hipSuccess)) {
            if (!lead) { lead = h; HIP_TRY(lead, hipSetDevice(lead->device)); }
            if (merge) {
                if (h->stream != lead->stream) HIP_TRY(h, hipStreamSynchronize(h->stream));   // (a member that took its level alone: its kernels ran on its own stream)
                RretryMerge a{};
                a.opt_list = h->opt_ptr; a.rlist = h->retry_list.as<int32_t>(); a.status = h->status.as<uint8_t>(); a.recd = h->recd.as<double>(); a.reci = h->reci.as<int32_t>();
                a.headd = h->headd.as<double>(); a.headi = h->headi.as<int32_t>(); a.epool = h->epool.as<double>();
                a.n_opt = (int)h->n_opt; a.n_rretry = (int)h->n_rretry; a.rec_d = (int)h->rec_d; a.rec_i = (int)h->rec_i; a.fd = h->fd; a.fi = h->fi; a.row0 = (int)h->n_erows;
                a.nx = h->n_x; a.nt = h->n_t; a.nc = h->n_c; a.ntc = h->n_tc; a.k = h->k;
                hipLaunchKernelGGL(k_rretry_merge, dim3(1), dim3(256), 0, lead->stream, a);
                HIP_TRY(h, hipGetLastError());
                h->rretry_rows = rows_out;
            }
            tab.push_back({h->headi.p, d_hi, (unsigned long long)h->n_opt * h->fi * sizeof(int32_t)});
            tab.push_back({h->headd.p, d_hd, (unsigned long long)h->n_opt * h->fd * sizeof(double)});
            if (rows_out > 0) tab.push_back({h->epool.p, d_er, (unsigned long long)rows_out * (h->n_t + 1) * sizeof(double)});
            n_slots[i] = h->n_opt; n_rows[i] = rows_out;
            continue;
        }
        (void)hipGetLastError();
        { static const bool dbg = [] { const char *ev = std::getenv("MPC_DEBUG_MANY"); return ev && ev[0] == '1'; }();
          if (dbg) std::fprintf(stderr, "[many] member %d outside the copy launch: level_done %d streamed %d regions %lld n_opt %lld region2 %d rretry %lld cap_slots %lld erows %lld cap_rows %lld\n", i, (int)h->level_done, (int)h->so.active,
                                (long long)h->n_regions, (long long)h->n_opt, (int)h->used_region2, (long long)h->n_rretry, (long long)cap_slots[i], (long long)h->n_erows, (long long)cap_rows[i]); }
        const int rc = mpc_level_regions_slots_nowait(h, head_d[i], head_i[i], cap_slots[i], erows[i], cap_rows[i], n_slots + i, n_rows + i);
        if (rc != MPC_OK) return rc;
    }
    if (!tab.empty()) {
        HIP_TRY(lead, hipSetDevice(lead->device));
        std::lock_guard<std::mutex> lk(g_fetch_mutex);
        if (g_fetch_event.count(lead->device) && g_fetch_event[lead->device]) HIP_TRY(lead, hipEventSynchronize(g_fetch_event[lead->device]));   // the table of the previous call is free again
        const size_t bytes = tab.size() * sizeof(FetchEntry);
        DevBuf &g_fetch_tab_dev = g_fetch_tab_dev_of[lead->device];
        HostBuf &g_fetch_tab_host = g_fetch_tab_host_of[lead->device];
        HIP_TRY(lead, g_fetch_tab_host.ensure(bytes));
        HIP_TRY(lead, g_fetch_tab_dev.ensure(bytes, lead->stream));
        std::memcpy(g_fetch_tab_host.p, tab.data(), bytes);
        HIP_TRY(lead, hipMemcpyAsync(g_fetch_tab_dev.p, g_fetch_tab_host.p, bytes, hipMemcpyHostToDevice, lead->stream));
        hipLaunchKernelGGL(k_fetch_many, dim3(8, (unsigned)tab.size()), dim3(256), 0, lead->stream, g_fetch_tab_dev.as<FetchEntry>());
        HIP_TRY(lead, hipGetLastError());
        auto &ev = g_fetch_event[lead->device];
        if (!ev) HIP_TRY(lead, pooled_event(&ev, false));
        HIP_TRY(lead, hipEventRecord(ev, lead->stream));
    }
    return MPC_OK;
}
int mpc_level_stream_fixup(mpc_handle *h, double *head_d, int32_t *head_i, double *erows, int64_t *n_rows) {
    if (!h || !head_d || !head_i || !erows) return MPC_ERR_INVALID;
    if (!h->so.active) return fail(h, MPC_ERR_STATE, "mpc_level_stream_fixup: this level was not streamed");
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "the level is still running"); }
    int64_t ns = 0;
    return level_regions_slots_impl(h, head_d, head_i, h->so.n_slots, erows, h->so.cap_rows, &ns, n_rows, false, true);
}
#ifdef MPC_XQ_HIST
// debug build only: histogram of k_xq's decisions, [outcome + 1][ratio tests passed]; reset after reading
int mpc_debug_xq_hist(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_xq_hist), 64 * sizeof(unsigned long long)) != hipSuccess) return MPC_ERR_HIP;
    if (hipMemcpyFromSymbol(out + 64, HIP_SYMBOL(g_xq_hist2), 64 * sizeof(unsigned long long)) != hipSuccess) return MPC_ERR_HIP;
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_xq_hist2), z, sizeof(z)) != hipSuccess) return MPC_ERR_HIP;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_xq_hist), z, sizeof(z)) == hipSuccess ? MPC_OK : MPC_ERR_HIP;
}
#endif
int mpc_engine_kind(mpc_handle *h, int32_t out[8]) {
    if (!h || !out) return MPC_ERR_INVALID;
    const bool fast = h->fast && !h->force_v1;
    out[0] = fast ? 1 : 0;
    out[1] = fast ? (h->fast_t & 1) + 1 : 0;
    out[2] = fast ? (h->fast_x & 1) + 1 : 0;
    out[3] = (fast && h->fast_r >= 0) ? (h->fast_r & 1) + 1 : 0;
    out[4] = fast ? (h->fast_t <= 1 ? 4 : (h->fast_t <= 3 ? 8 : 10)) : 0;
    out[5] = h->theta_open ? 1 : 0;
    out[6] = h->kkt_mode;
    out[7] = (fast && h->kkt_mode == 0 && h->no_kkt_thread != 1) ? KKT_THREAD_MAX : 0;
    return MPC_OK;
}
int mpc_sync(mpc_handle *h) {
    if (!h) return MPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (fetch_many_wait(h->device) != MPC_OK) return fail(h, MPC_ERR_HIP, "mpc_sync: the shared record copy (k_fetch_many) failed");   // (its stream may be another member's)
    return MPC_OK;
}

// ---- pooled page-locked host memory (host_pool_* above) ------------------------------------------------------------------
int mpc_host_alloc(uint64_t bytes, void **out) {
    if (!out) return MPC_ERR_INVALID;
    *out = nullptr;
    if (host_pool_take((size_t)bytes, out, nullptr) != hipSuccess) return fail(nullptr, MPC_ERR_HIP, "hipHostMalloc failed");
    return MPC_OK;
}

int mpc_host_free(void *p) {
    if (!p) return MPC_OK;
    return host_pool_give(p) ? MPC_OK : MPC_ERR_INVALID;
}

int mpc_level_regions(mpc_handle *h, double *rec_d, int32_t *rec_i, int64_t *cand_index, int64_t cap) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (cap < h->n_regions) return fail(h, MPC_ERR_CAPACITY, "region buffer too small");
    if (h->n_regions == 0) return MPC_OK;
    if (!rec_d || !rec_i) return MPC_ERR_INVALID;
    // fetch in compact form, then expand into the fixed-stride records of the header
    const int nx = h->n_x, nt = h->n_t, nc = h->n_c, ntc = h->n_tc, k = h->k, nr = nt + 1, fd = h->fd, fi = h->fi;
    int64_t max_rows = 0, nreg = 0, nrows = 0;
    mpc_compact_strides(h, nullptr, nullptr, &max_rows);
    std::vector<double> c_hd((size_t)h->n_regions * fd), c_er((size_t)std::max<int64_t>(max_rows, 1) * nr);
    std::vector<int32_t> c_hi((size_t)h->n_regions * fi);
    int rc = mpc_level_regions_compact(h, c_hd.data(), c_hi.data(), h->n_regions, c_er.data(), max_rows, &nreg, &nrows);
    if (rc) return rc;
    for (int64_t w = 0; w < nreg; ++w) {
        double *rd = rec_d + w * h->rec_d;
        int32_t *ri = rec_i + w * h->rec_i;
        std::fill(rd, rd + h->rec_d, 0.0);
        std::fill(ri, ri + h->rec_i, -1);
        const double *sd = c_hd.data() + w * fd;
        const int32_t *si = c_hi.data() + w * fi;
        const int nE = si[2], n_om = si[3], n_la = si[4], n_re = si[5], off = si[6];
        std::memcpy(rd, sd, sizeof(double) * (nx * nt + nx));
        double *Al = rd + nx * nt + nx, *bl = Al + (size_t)nc * nt, *E = bl + nc, *f = E + (size_t)(nc + ntc) * nt;
        std::memcpy(Al, sd + nx * nt + nx, sizeof(double) * k * nt);
        std::memcpy(bl, sd + nx * nt + nx + k * nt, sizeof(double) * k);
        for (int r = 0; r < nE; ++r) {
            const double *row = c_er.data() + (size_t)(off + r) * nr;
            f[r] = row[0];
            for (int t = 0; t < nt; ++t) E[(size_t)r * nt + t] = row[1 + t];
        }
        ri[0] = k; ri[1] = nE; ri[2] = n_om; ri[3] = n_la; ri[4] = n_re;
        const int32_t *act = si + 8, *om = act + k, *la = om + ntc, *ridx = la + k, *rcon = ridx + (nc - k);
        std::memcpy(ri + 5, act, sizeof(int32_t) * k);
        std::memcpy(ri + 5 + nc, om, sizeof(int32_t) * n_om);
        std::memcpy(ri + 5 + nc + ntc, la, sizeof(int32_t) * n_la);
        std::memcpy(ri + 5 + nc + ntc + nc, ridx, sizeof(int32_t) * n_re);
        std::memcpy(ri + 5 + nc + ntc + nc + nc, rcon, sizeof(int32_t) * n_re);
        if (cand_index) cand_index[w] = si[1];
    }
    return MPC_OK;
}

static int copy_out(mpc_handle *h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (bytes == 0) return MPC_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, kind, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // caller-owned memory: complete before returning
    return MPC_OK;
}

int mpc_level_children(mpc_handle *h, int32_t *out, int64_t cap) {
    if (!h || (!out && cap > 0)) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (cap < h->n_children) return fail(h, MPC_ERR_CAPACITY, "children buffer too small");
    return copy_out(h, out, h->children.p, (size_t)h->n_children * (h->k + 1) * sizeof(int32_t), hipMemcpyDeviceToHost);
}
int mpc_level_children_device(mpc_handle *h, int32_t *out, int64_t cap) {
    if (!h || (!out && cap > 0)) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (cap < h->n_children) return fail(h, MPC_ERR_CAPACITY, "children buffer too small");
    return copy_out(h, out, h->children.p, (size_t)h->n_children * (h->k + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice);
}
int mpc_level_pruned_new(mpc_handle *h, uint64_t *out, int64_t cap) {
    if (!h || (!out && cap > 0)) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (cap < h->n_pruned_new) return fail(h, MPC_ERR_CAPACITY, "pruned buffer too small");
    return copy_out(h, out, h->pruned.as<uint64_t>() + (size_t)h->n_pruned * h->mw, (size_t)h->n_pruned_new * h->mw * sizeof(uint64_t), hipMemcpyDeviceToHost);
}
int mpc_level_regions_device(mpc_handle *h, double *head_d_dev, int32_t *head_i_dev, double *erows_dev, int64_t cap_slots,
                             int64_t cap_rows, int64_t *n_slots, int64_t *n_rows) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (n_slots) *n_slots = 0;
    if (n_rows) *n_rows = 0;
    if (h->n_regions == 0 || h->n_opt == 0) return MPC_OK;
    // only the common case lives entirely on the device: every region came from k_region2; otherwise the caller takes the
    // host route (mpc_level_regions_slots)
    if (!h->used_region2 || h->n_rretry > 0 || h->so.active) return fail(h, MPC_ERR_STATE, "records of this level are not all in slot form on the device");
    if (cap_slots < h->n_opt || cap_rows < h->n_erows) return fail(h, MPC_ERR_CAPACITY, "slot / row buffers too small");
    if (!head_d_dev || !head_i_dev || (!erows_dev && h->n_erows > 0)) return MPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    HIP_TRY(h, hipMemcpyAsync(head_d_dev, h->headd.p, (size_t)h->n_opt * h->fd * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(head_i_dev, h->headi.p, (size_t)h->n_opt * h->fi * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (h->n_erows > 0) HIP_TRY(h, hipMemcpyAsync(erows_dev, h->epool.p, (size_t)h->n_erows * (h->n_t + 1) * sizeof(double), hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (n_slots) *n_slots = h->n_opt;
    if (n_rows) *n_rows = h->n_erows;
    return MPC_OK;
}

int mpc_level_pruned_new_device(mpc_handle *h, uint64_t *out, int64_t cap) {
    if (!h || (!out && cap > 0)) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    if (cap < h->n_pruned_new) return fail(h, MPC_ERR_CAPACITY, "pruned buffer too small");
    return copy_out(h, out, h->pruned.as<uint64_t>() + (size_t)h->n_pruned * h->mw, (size_t)h->n_pruned_new * h->mw * sizeof(uint64_t), hipMemcpyDeviceToDevice);
}

int mpc_frontier_advance(mpc_handle *h) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->level_done) return fail(h, MPC_ERR_STATE, "mpc_level_run has not been called for this frontier");
    h->n_pruned += h->n_pruned_new + h->n_pruned_extra;   // already in place behind the list (no copy)
    h->n_pruned_extra = 0;
    std::swap(h->frontier, h->children);
    std::swap(h->parent_slot, h->parent_slot_next);
    h->have_parent_slot = true;
    h->have_prev_dict = h->storing;
    h->dict_cur = 1 - h->dict_cur;
    h->storing = false;
    h->prev_regions = h->n_regions;
    h->n_prev = h->n;   // the frontier that has just been left sits in h->children until this level writes its own children
    h->n = h->n_children;
    h->k = h->k + 1;
    h->level_done = false;
    h->n_children = 0; h->n_pruned_new = 0; h->n_regions = 0; h->n_opt = 0;
    return MPC_OK;
}
int mpc_frontier_advance_batch(mpc_handle **hs, int32_t n_handles) {
    if (!hs || n_handles < 0) return MPC_ERR_INVALID;
    for (int i = 0; i < n_handles; ++i) { const int rc = mpc_frontier_advance(hs[i]); if (rc != MPC_OK) return rc; }
    return MPC_OK;
}

// ---- the level loop of many programs on a thread of the library (include/mpcombi.h: mpc_solve_many_*) ----------------------------------
struct ManyLevel {
    std::vector<int32_t> member;
    std::vector<mpc_level_stats> stats;
    std::vector<int64_t> ns, nr, od, oi, oe;
    void *hd = nullptr, *hi = nullptr, *er = nullptr;
    int64_t ld = 0, li = 0, le = 0;
    int32_t n_shared = 0;
    double ms_wall = 0;
    bool base = false;       // the closing level of the base active sets
    bool taken = false;      // the blocks belong to the caller
    ~ManyLevel() { if (!taken) { if (hd) (void)host_pool_give(hd); if (hi) (void)host_pool_give(hi); if (er) (void)host_pool_give(er); } }
};
struct ManyJob {
    std::vector<mpc_handle *> hs;
    std::vector<int32_t> max_levels;
    int32_t flags = 0;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::unique_ptr<ManyLevel>> levels;   // appended by the loop, read by the caller (under m)
    bool finished = false, handover = false;
    int rc = MPC_OK;
};

static void many_loop(ManyJob *J) {
    auto finish = [&](int rc) {
        std::lock_guard<std::mutex> lk(J->m);
        J->rc = rc; J->finished = true;
        J->cv.notify_all();
    };
    const int B = (int)J->hs.size();
    if (B == 0) return finish(MPC_OK);
    mpc_handle *h0 = J->hs[0];
    if (hipSetDevice(h0->device) != hipSuccess) return finish(fail(h0, MPC_ERR_HIP, "mpc_solve_many: hipSetDevice failed"));
    const double budget_gb = 0.6 * [] { const char *ev = std::getenv("MPC_BATCH_BUDGET_GB"); return ev ? std::atof(ev) : 160.0; }();   // (headroom as in the host layer's loop)
    std::vector<int> depth(B, 0);
    std::vector<int32_t> active;
    for (int i = 0; i < B; ++i) {
        mpc_handle *h = J->hs[i];
        int rc = mpc_pruned_clear(h);
        if (rc == MPC_OK) rc = mpc_frontier_root(h);
        if (rc != MPC_OK) return finish(rc);
        if (J->max_levels[i] > 0) active.push_back(i);
    }
    auto gen_of = [&](int i) { return depth[i] + 1 != J->max_levels[i] ? 1 : 0; };
    auto start = [&](const std::vector<int32_t> &act, void **token) -> int {
        std::vector<mpc_handle *> hh; std::vector<int32_t> gg;
        for (int i : act) { hh.push_back(J->hs[i]); gg.push_back(gen_of(i)); }
        return mpc_level_batch_start(hh.data(), (int32_t)hh.size(), gg.data(), J->flags & ~MPC_SOLVE_MANY_BASE, token);
    };
    void *token = nullptr;
    static const bool dbg = [] { const char *ev = std::getenv("MPC_DEBUG_MANY"); return ev && ev[0] == '1'; }();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(now() - a).count(); };
    { const auto ts = now(); if (!active.empty()) { const int rc = start(active, &token); if (rc != MPC_OK) return finish(rc); } if (dbg) std::fprintf(stderr, "[many] first start %.3f ms\n", ms_since(ts)); }
    bool base_phase = false;
    auto start_base = [&]() -> int {      // every program's base active set as one more shared level (reference driver :142-146)
        active.clear();
        for (int i = 0; i < B; ++i) {
            mpc_handle *h = J->hs[i];
            int rc = frontier_reset(h, 1, h->n_eq);
            if (rc != MPC_OK) return rc;
            if (h->n_eq > 0) { hipLaunchKernelGGL(k_base_frontier, dim3(1), dim3(64), 0, h->stream, h->n_eq, h->frontier.as<int32_t>()); HIP_TRY(h, hipGetLastError()); }
            h->n_pruned = 0; h->n_pruned_extra = 0;
            active.push_back(i);
        }
        for (int i = 0; i < B; ++i) if (J->hs[i]->stream != J->hs[0]->stream && J->hs[i]->n_eq > 0) HIP_TRY(J->hs[i], hipStreamSynchronize(J->hs[i]->stream));   // (the shared launches run on the first member's stream)
        base_phase = true;
        std::vector<mpc_handle *> hh(J->hs); std::vector<int32_t> gg(B, 0);
        return mpc_level_batch_start(hh.data(), B, gg.data(), J->flags & ~MPC_SOLVE_MANY_BASE, &token);
    };
    if (active.empty() && (J->flags & MPC_SOLVE_MANY_BASE)) { const int rc = start_base(); if (rc != MPC_OK) return finish(rc); }
    while (!active.empty()) {
        const auto t0 = std::chrono::steady_clock::now();
        double t_wait = 0, t_fetch = 0, t_adv = 0, t_start = 0;
        const int nb = (int)active.size();
        std::unique_ptr<ManyLevel> L(new ManyLevel());
        L->member = active;
        L->base = base_phase;
        L->stats.resize(nb);
        {
            // (mpc_level_batch_wait writes stats[m.id] with ids = positions in ITS handle list = positions in `active`)
            const auto ts = now();
            const int rc = mpc_level_batch_wait(token, L->stats.data(), &L->n_shared);
            t_wait = ms_since(ts);
            token = nullptr;
            if (rc != MPC_OK) return finish(rc);
        }
        // the level's records: three blocks for all members (sizes from each member's statistics, as the host layer's level_batch_fetch)
        const auto tf = now();
        L->ns.assign(nb, 0); L->nr.assign(nb, 0); L->od.assign(nb, 0); L->oi.assign(nb, 0); L->oe.assign(nb, 0);
        std::vector<int64_t> fds(nb), fis(nb);
        for (int j = 0; j < nb; ++j) {
            const mpc_handle *h = J->hs[active[j]];
            const mpc_level_stats &st = L->stats[j];
            const int k = st.k;
            fds[j] = (int64_t)h->n_x * h->n_t + h->n_x + (int64_t)k * h->n_t + k;
            fis[j] = 8 + k + h->n_tc + k + 2 * (h->n_c - k);
            const int64_t rows_t = h->n_c - h->n_eq + h->n_tc;
            if (st.n_regions > 0) {
                L->ns[j] = st.n_opt;
                L->nr[j] = st.n_region_rows ? st.n_region_rows + st.n_region_retry * rows_t : st.n_regions * rows_t;
            }
            L->od[j] = L->ld; L->oi[j] = L->li; L->oe[j] = L->le;
            L->ld += L->ns[j] * fds[j]; L->li += L->ns[j] * fis[j]; L->le += L->nr[j] * (h->n_t + 1);
        }
        if (L->ld > 0) {
            if (host_pool_take((size_t)L->ld * 8, &L->hd, nullptr) != hipSuccess || host_pool_take((size_t)L->li * 4, &L->hi, nullptr) != hipSuccess ||
                host_pool_take((size_t)std::max<int64_t>(L->le, 1) * 8, &L->er, nullptr) != hipSuccess)
                return finish(fail(h0, MPC_ERR_HIP, "mpc_solve_many: page-locked memory for the level's records"));
            std::vector<mpc_handle *> hh; std::vector<double *> pd, pe; std::vector<int32_t *> pi; std::vector<int64_t> cs, cr, n1, n2; std::vector<int> at;
            for (int j = 0; j < nb; ++j) {
                if (L->ns[j] == 0) continue;
                hh.push_back(J->hs[active[j]]); at.push_back(j);
                pd.push_back(static_cast<double *>(L->hd) + L->od[j]); pi.push_back(static_cast<int32_t *>(L->hi) + L->oi[j]);
                pe.push_back(static_cast<double *>(L->er) + L->oe[j]);
                cs.push_back(L->ns[j]); cr.push_back(L->nr[j]);
            }
            n1.assign(hh.size(), 0); n2.assign(hh.size(), 0);
            const int rc = mpc_level_batch_fetch(hh.data(), (int32_t)hh.size(), pd.data(), pi.data(), cs.data(), pe.data(), cr.data(), n1.data(), n2.data());
            if (rc != MPC_OK) return finish(rc);
            for (size_t q = 0; q < hh.size(); ++q) { L->ns[at[q]] = n1[q]; L->nr[at[q]] = n2[q]; }
            // a member outside the shared copy launch (records re-solved by the LDS engine, streamed ...) queued its copies on its own stream
            for (mpc_handle *h : hh) if (hipStreamQuery(h->stream) != hipSuccess) { (void)hipGetLastError(); if (hipStreamSynchronize(h->stream) != hipSuccess) return finish(fail(h, MPC_ERR_HIP, "mpc_solve_many: record copy")); }
        }
        t_fetch = ms_since(tf);
        const auto ta = now();
        std::vector<int32_t> nxt;
        if (!base_phase) for (int j = 0; j < nb; ++j) if (gen_of(active[j]) && L->stats[j].n_children > 0) nxt.push_back(active[j]);
        double need_gb = 0.0;
        for (int i : nxt) {
            const int rc = mpc_frontier_advance(J->hs[i]);
            if (rc != MPC_OK) return finish(rc);
            depth[i] += 1;
            need_gb += batch_level_gb(J->hs[i], gen_of(i));
        }
        const bool handover = !nxt.empty() && need_gb > budget_gb;
        t_adv = ms_since(ta);
        const auto tst = now();
        const bool then_base = nxt.empty() && !base_phase && (J->flags & MPC_SOLVE_MANY_BASE);
        if (!nxt.empty() && !handover) {
            const int rc = start(nxt, &token);      // (waits for the shared record copy first: the members' record buffers are written again)
            if (rc != MPC_OK) return finish(rc);
        } else if (then_base) {
            const int rc = start_base();
            if (rc != MPC_OK) return finish(rc);
        } else if (fetch_many_wait(h0->device) != MPC_OK) return finish(fail(h0, MPC_ERR_HIP, "mpc_solve_many: the shared record copy failed"));
        t_start = ms_since(tst);
        L->ms_wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (dbg) std::fprintf(stderr, "[many] level of %d members: wait+finish %.3f fetch %.3f advance %.3f next start %.3f total %.3f ms\n", nb, t_wait, t_fetch, t_adv, t_start, L->ms_wall);
        {
            std::lock_guard<std::mutex> lk(J->m);
            J->levels.push_back(std::move(L));
            if (handover) J->handover = true;
            J->cv.notify_all();
        }
        if (handover) break;
        if (then_base) continue;      // (`active` = every member, set by start_base)
        active.swap(nxt);
    }
    finish(MPC_OK);
}

extern "C" int mpc_solve_many_start(mpc_handle **hs, int32_t n_handles, const int32_t *max_levels, int32_t flags, void **job) {
    if (!hs || n_handles < 0 || !max_levels || !job) return MPC_ERR_INVALID;
    *job = nullptr;
    for (int i = 0; i < n_handles; ++i) {
        if (!hs[i]) return MPC_ERR_INVALID;
        std::lock_guard<std::mutex> lk(hs[i]->wm);
        if (hs[i]->w_busy) return fail(hs[i], MPC_ERR_STATE, "a level started with mpc_level_start is still running");
    }
    std::unique_ptr<ManyJob> J(new ManyJob());
    J->hs.assign(hs, hs + n_handles);
    J->max_levels.assign(max_levels, max_levels + n_handles);
    J->flags = flags & (MPC_LEVEL_KEEP_LOWDIM | MPC_SOLVE_MANY_BASE);
    ManyJob *raw = J.get();
    try { J->th = std::thread(many_loop, raw); } catch (...) { return fail(n_handles ? hs[0] : nullptr, MPC_ERR_STATE, "mpc_solve_many_start: no thread"); }
    *job = J.release();
    return MPC_OK;
}
extern "C" int mpc_solve_many_level(void *job, int32_t level, mpc_many_level_info *info) {
    if (!job || !info || level < 0) return MPC_ERR_INVALID;
    ManyJob *J = static_cast<ManyJob *>(job);
    std::unique_lock<std::mutex> lk(J->m);
    J->cv.wait(lk, [&] { return (int)J->levels.size() > level || J->finished; });
    std::memset(info, 0, sizeof(*info));
    info->level = level;
    if ((int)J->levels.size() <= level) {
        info->done = J->handover ? 2 : 1;
        return J->rc;
    }
    ManyLevel &L = *J->levels[level];
    info->n_members = (int32_t)L.member.size(); info->n_shared = L.n_shared; info->base = L.base ? 1 : 0;
    info->member = L.member.data(); info->stats = L.stats.data();
    info->n_slots = L.ns.data(); info->n_rows = L.nr.data(); info->off_d = L.od.data(); info->off_i = L.oi.data(); info->off_e = L.oe.data();
    info->head_d = static_cast<double *>(L.hd); info->head_i = static_cast<int32_t *>(L.hi); info->erows = static_cast<double *>(L.er);
    info->len_d = L.ld; info->len_i = L.li; info->len_e = std::max<int64_t>(L.le, L.hd ? 1 : 0);
    info->ms_wall = L.ms_wall;
    L.taken = true;
    return MPC_OK;
}
extern "C" int mpc_solve_many_wait(void *job) {
    if (!job) return MPC_ERR_INVALID;
    std::unique_ptr<ManyJob> J(static_cast<ManyJob *>(job));
    if (J->th.joinable()) J->th.join();
    return J->rc;
}

}  // extern "C"

// ---- connected-graph traversal, bookkeeping on the device (graph.hpp) ---------------------------------------------------------
namespace {
struct GDec2 { __host__ __device__ rocprim::tuple<unsigned long long &, unsigned long long &> operator()(GMask<2> &k) const {
    return rocprim::tuple<unsigned long long &, unsigned long long &>(k.w[1], k.w[0]); } };
struct GDec4 { __host__ __device__ rocprim::tuple<unsigned long long &, unsigned long long &, unsigned long long &, unsigned long long &> operator()(GMask<4> &k) const {
    return rocprim::tuple<unsigned long long &, unsigned long long &, unsigned long long &, unsigned long long &>(k.w[3], k.w[2], k.w[1], k.w[0]); } };
template <int MW> struct GDec;
template <> struct GDec<2> { typedef GDec2 type; };
template <> struct GDec<4> { typedef GDec4 type; };
constexpr long long G_CHUNK = 1ll << 22;   // masks per group run (bounds the level buffers and the 32-bit neighbour offsets)

template <int MW>
int g_sort(mpc_handle *h, const GMask<MW> *in, GMask<MW> *out, long long n) {
    if (n <= 0) return MPC_OK;
    size_t bytes = 0;
    typename GDec<MW>::type dec;
    HIP_TRY(h, rocprim::radix_sort_keys(nullptr, bytes, const_cast<GMask<MW> *>(in), out, (size_t)n, dec, 0u, 64u * MW, h->stream));
    HIP_TRY(h, h->g.sort_tmp.ensure(std::max<size_t>(bytes, 16), h->stream));
    bytes = h->g.sort_tmp.cap;
    HIP_TRY(h, rocprim::radix_sort_keys(h->g.sort_tmp.p, bytes, const_cast<GMask<MW> *>(in), out, (size_t)n, dec, 0u, 64u * MW, h->stream));
    return MPC_OK;
}

// pending (n_pending masks, any order, duplicates) -> next wave (sorted by cardinality, then mask) and visited := visited + new
template <int MW>
int g_close(mpc_handle *h) {
    auto &g = h->g;
    hipStream_t st = h->stream;
    const long long n = g.n_pending;
    g.gk.clear(); g.goff.clear(); g.gcnt.clear();
    g.n_wave = 0;
    g.n_pending = 0;
    if (n == 0) return MPC_OK;
    typedef GMask<MW> M;
    const unsigned nb = (unsigned)((n + 255) / 256);
    HIP_TRY(h, g.tmp_a.ensure((size_t)n * sizeof(M), st));
    { int rc = g_sort<MW>(h, g.pending.as<M>(), g.tmp_a.as<M>(), n); if (rc) return rc; }
    HIP_TRY(h, h->flag.ensure((size_t)n * sizeof(int32_t), st));
    HIP_TRY(h, h->pos.ensure((size_t)n * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_g_newflags<MW>, dim3(nb), dim3(256), 0, st, g.tmp_a.as<M>(), n, g.visited.as<M>(), g.n_visited, h->flag.as<int32_t>());
    { int rc = launch_scan(h, h->flag.as<int32_t>(), h->pos.as<int32_t>(), n, h->tot_dev); if (rc) return rc; }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(st));
    const long long n_new = h->tot_host[0];
    if (n_new == 0) return MPC_OK;
    // visited ++ new, sorted
    HIP_TRY(h, g.tmp_b.ensure((size_t)(g.n_visited + n_new) * sizeof(M), st));
    if (g.n_visited > 0) HIP_TRY(h, hipMemcpyAsync(g.tmp_b.p, g.visited.p, (size_t)g.n_visited * sizeof(M), hipMemcpyDeviceToDevice, st));
    M *fresh = g.tmp_b.as<M>() + g.n_visited;   // the new masks, sorted by mask
    hipLaunchKernelGGL(k_g_compact<MW>, dim3(nb), dim3(256), 0, st, g.tmp_a.as<M>(), n, h->flag.as<int32_t>(), h->pos.as<int32_t>(), fresh);
    HIP_TRY(h, hipGetLastError());
    // next wave: the new masks in (cardinality, mask) order -- a stable sort of their cardinalities carries the permutation
    const unsigned nbn = (unsigned)((n_new + 255) / 256);
    for (DevBuf *b : {&g.card, &g.idx, &g.card2, &g.idx2}) HIP_TRY(h, b->ensure((size_t)n_new * sizeof(unsigned int), st));
    hipLaunchKernelGGL(k_g_card<MW>, dim3(nbn), dim3(256), 0, st, fresh, n_new, g.card.as<unsigned int>(), g.idx.as<unsigned int>());
    {
        size_t bytes = 0;
        HIP_TRY(h, rocprim::radix_sort_pairs(nullptr, bytes, g.card.as<unsigned int>(), g.card2.as<unsigned int>(), g.idx.as<unsigned int>(), g.idx2.as<unsigned int>(),
                                             (size_t)n_new, 0u, 9u, st));
        HIP_TRY(h, g.sort_tmp.ensure(std::max<size_t>(bytes, 16), st));
        bytes = g.sort_tmp.cap;
        HIP_TRY(h, rocprim::radix_sort_pairs(g.sort_tmp.p, bytes, g.card.as<unsigned int>(), g.card2.as<unsigned int>(), g.idx.as<unsigned int>(), g.idx2.as<unsigned int>(),
                                             (size_t)n_new, 0u, 9u, st));
    }
    HIP_TRY(h, g.wave.ensure((size_t)n_new * sizeof(M), st));
    hipLaunchKernelGGL(k_g_gather<MW>, dim3(nbn), dim3(256), 0, st, fresh, g.idx2.as<unsigned int>(), n_new, g.wave.as<M>());
    HIP_TRY(h, g.hist.ensure(257 * sizeof(int32_t), st));
    HIP_TRY(h, hipMemsetAsync(g.hist.p, 0xff, 257 * sizeof(int32_t), st));   // -1: no mask of that cardinality
    hipLaunchKernelGGL(k_g_first, dim3(nbn), dim3(256), 0, st, g.card2.as<unsigned int>(), n_new, g.hist.as<int32_t>());
    HIP_TRY(h, hipGetLastError());
    int32_t hist[257];
    HIP_TRY(h, hipMemcpyAsync(hist, g.hist.p, sizeof(hist), hipMemcpyDeviceToHost, st));
    // visited := sort(visited ++ new)   (tmp_b -> visited)
    HIP_TRY(h, g.visited.ensure((size_t)(g.n_visited + n_new) * sizeof(M), st));
    { int rc = g_sort<MW>(h, g.tmp_b.as<M>(), g.visited.as<M>(), g.n_visited + n_new); if (rc) return rc; }
    HIP_TRY(h, hipStreamSynchronize(st));
    g.n_visited += n_new;
    g.n_wave = n_new;
    // hist[k] = index of the first mask with k rows (-1: none); a group ends where the next present cardinality starts
    long long end = n_new;
    std::vector<std::pair<int, std::pair<long long, long long>>> groups;   // (k, (first, count)), built from the back
    for (int k = 256; k >= 0; --k) {
        if (hist[k] < 0) continue;
        groups.push_back({k, {hist[k], end - hist[k]}});
        end = hist[k];
    }
    for (auto it = groups.rbegin(); it != groups.rend(); ++it) {
        long long off = it->second.first, c = it->second.second;
        while (c > 0) { const long long take = std::min(c, G_CHUNK); g.gk.push_back(it->first); g.goff.push_back(off); g.gcnt.push_back(take); off += take; c -= take; }
    }
    return MPC_OK;
}

template <int MW>
int g_group_run(mpc_handle *h, int gi, mpc_level_stats *stats) {
    auto &g = h->g;
    typedef GMask<MW> M;
    hipStream_t st = h->stream;
    const int k = g.gk[gi];
    const long long off = g.goff[gi], cnt = g.gcnt[gi];
    const M *masks = g.wave.as<M>() + off;
    const unsigned nb = (unsigned)((cnt + 255) / 256);
    const bool too_many_rows = k > std::min(h->n_x, h->n_c);   // more rows than variables: rank deficient by counting (is_full_rank)
    mpc_level_stats ls;
    std::memset(&ls, 0, sizeof(ls));
    ls.n = cnt; ls.k = k;
    if (too_many_rows) ls.n_status[ST_INFEASIBLE] = cnt;
    else {
        int rc = frontier_reset(h, cnt, k);
        if (rc) return rc;
        if (k > 0) hipLaunchKernelGGL(k_g_frontier<MW>, dim3(nb), dim3(256), 0, st, masks, cnt, k, h->frontier.as<int32_t>());
        HIP_TRY(h, hipGetLastError());
        // Both traversals only need "rank deficient / no region / non-empty but lower dimensional / region": mpqp_graph.py hands on
        // the same subsets whether a set is infeasible or feasible but not optimal (:69-91; only its pruning list tells them apart,
        // and no pruning list is kept here), so the (x,theta) feasibility LP is left out for it as well.
        rc = level_run_impl(h, 0, MPC_LEVEL_GRAPH, &ls);
        if (rc) return rc;
    }
    if (stats) *stats = ls;
    // facet constraints of the regions (variant 1)
    const M *facet = nullptr;
    if (g.variant == 1 && !too_many_rows && ls.n_regions > 0) {
        HIP_TRY(h, g.facet.ensure((size_t)cnt * sizeof(M), st));
        HIP_TRY(h, hipMemsetAsync(g.facet.p, 0, (size_t)cnt * sizeof(M), st));
        if (h->used_region2 && h->n_opt > 0)
            hipLaunchKernelGGL(k_g_facets_slots<MW>, dim3((unsigned)((h->n_opt + 255) / 256)), dim3(256), 0, st, h->headi.as<int32_t>(), h->fi, h->n_opt, k, h->n_c,
                               h->n_tc, g.facet.as<M>());
        const int32_t *list = h->used_region2 ? h->retry_list.as<int32_t>() : h->opt_ptr;
        const long long n_list = h->used_region2 ? h->n_rretry : h->n_opt;
        if (n_list > 0)
            hipLaunchKernelGGL(k_g_facets_fixed<MW>, dim3((unsigned)((n_list + 255) / 256)), dim3(256), 0, st, h->reci.as<int32_t>(), h->rec_i, list, n_list,
                               h->status.as<uint8_t>(), h->n_c, h->n_tc, g.facet.as<M>());
        HIP_TRY(h, hipGetLastError());
        facet = g.facet.as<M>();
    } else if (g.variant == 1) {
        HIP_TRY(h, g.facet.ensure((size_t)cnt * sizeof(M), st));
        HIP_TRY(h, hipMemsetAsync(g.facet.p, 0, (size_t)cnt * sizeof(M), st));
        facet = g.facet.as<M>();
    }
    // neighbours: count -> scan -> emit behind what the earlier groups of this wave have emitted
    M eq, all;
    for (int j = 0; j < MW; ++j) { eq.w[j] = 0; all.w[j] = 0; }
    for (int i = 0; i < h->n_eq; ++i) eq.w[i >> 6] |= 1ull << (i & 63);
    for (int i = 0; i < h->n_c; ++i) all.w[i >> 6] |= 1ull << (i & 63);
    const uint8_t *status = too_many_rows ? nullptr : h->status.as<uint8_t>();
    HIP_TRY(h, g.cnt.ensure((size_t)cnt * sizeof(int32_t), st));
    HIP_TRY(h, g.off.ensure((size_t)cnt * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_g_count<MW>, dim3(nb), dim3(256), 0, st, masks, cnt, status, (int)ST_INFEASIBLE, g.variant, eq, all, facet, g.cnt.as<int32_t>());
    { int rc = launch_scan(h, g.cnt.as<int32_t>(), g.off.as<int32_t>(), cnt, h->tot_dev); if (rc) return rc; }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(st));
    const long long total = h->tot_host[0];
    if (total > 0) {
        HIP_TRY(h, g.pending.ensure((size_t)(g.n_pending + total) * sizeof(M), st, true));
        hipLaunchKernelGGL(k_g_emit<MW>, dim3(nb), dim3(256), 0, st, masks, cnt, status, (int)ST_INFEASIBLE, g.variant, eq, all, facet, g.off.as<int32_t>(),
                           g.pending.as<M>() + g.n_pending);
        HIP_TRY(h, hipGetLastError());
        g.n_pending += total;
    }
    return MPC_OK;
}
}  // namespace

extern "C" {

int mpc_graph_begin(mpc_handle *h, const uint64_t *seed_masks, int64_t n_seeds, int32_t variant) {
    if (!h || (n_seeds > 0 && !seed_masks) || n_seeds < 0 || variant < 0 || variant > 1) return MPC_ERR_INVALID;
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "a level started with mpc_level_start is still running"); }
    HIP_TRY(h, hipSetDevice(h->device));
    auto &g = h->g;
    g.active = true; g.variant = variant; g.n_wave = g.n_visited = 0; g.n_pending = n_seeds;
    if (n_seeds > 0) {
        HIP_TRY(h, g.pending.ensure((size_t)n_seeds * h->mw * sizeof(uint64_t), h->stream));
        HIP_TRY(h, hipMemcpyAsync(g.pending.p, seed_masks, (size_t)n_seeds * h->mw * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return h->mw == 2 ? g_close<2>(h) : g_close<4>(h);
}

int mpc_graph_wave(mpc_handle *h, int32_t *k_list, int64_t *count_list, int32_t cap, int32_t *n_groups, int64_t *n_wave, int64_t *n_visited) {
    if (!h || !n_groups) return MPC_ERR_INVALID;
    if (!h->g.active) return fail(h, MPC_ERR_STATE, "mpc_graph_begin has not been called");
    const int ng = (int)h->g.gk.size();
    *n_groups = ng;
    if (n_wave) *n_wave = h->g.n_wave;
    if (n_visited) *n_visited = h->g.n_visited;
    if (cap < ng) return ng == 0 ? MPC_OK : fail(h, MPC_ERR_CAPACITY, "group arrays too small");
    for (int i = 0; i < ng; ++i) { if (k_list) k_list[i] = h->g.gk[i]; if (count_list) count_list[i] = h->g.gcnt[i]; }
    return MPC_OK;
}

int mpc_graph_group_run(mpc_handle *h, int32_t group, mpc_level_stats *stats) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->g.active || group < 0 || group >= (int)h->g.gk.size()) return fail(h, MPC_ERR_STATE, "no such group in the current wave");
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "a level started with mpc_level_start is still running"); }
    HIP_TRY(h, hipSetDevice(h->device));
    return h->mw == 2 ? g_group_run<2>(h, group, stats) : g_group_run<4>(h, group, stats);
}

int mpc_graph_wave_close(mpc_handle *h, int64_t *n_next, int64_t *n_visited) {
    if (!h) return MPC_ERR_INVALID;
    if (!h->g.active) return fail(h, MPC_ERR_STATE, "mpc_graph_begin has not been called");
    HIP_TRY(h, hipSetDevice(h->device));
    const int rc = h->mw == 2 ? g_close<2>(h) : g_close<4>(h);
    if (n_next) *n_next = h->g.n_wave;
    if (n_visited) *n_visited = h->g.n_visited;
    return rc;
}

int mpc_check_level(mpc_handle *h, const int32_t *cand, int64_t n, int32_t k, const uint64_t *pruned_masks, int64_t m,
                    int32_t gen_children, uint8_t *status, int64_t *n_regions, double *rec_d, int32_t *rec_i,
                    int64_t *region_cand, int64_t region_cap, int64_t *n_children, int32_t *children, int64_t children_cap) {
    if (!h) return MPC_ERR_INVALID;
    int rc = mpc_frontier_set(h, cand, n, k);
    if (rc) return rc;
    mpc_pruned_clear(h);
    rc = mpc_pruned_add(h, pruned_masks, m);
    if (rc) return rc;
    mpc_level_stats stats;
    rc = mpc_level_run(h, gen_children, &stats);
    if (rc) return rc;
    if (status) { rc = mpc_level_status(h, status); if (rc) return rc; }
    if (n_regions) *n_regions = stats.n_regions;
    if (n_children) *n_children = stats.n_children;
    if (stats.n_regions > region_cap || stats.n_children > children_cap) return fail(h, MPC_ERR_CAPACITY, "output buffers too small");
    if (stats.n_regions > 0) { rc = mpc_level_regions(h, rec_d, rec_i, region_cand, region_cap); if (rc) return rc; }
    if (stats.n_children > 0) { rc = mpc_level_children(h, children, children_cap); if (rc) return rc; }
    return MPC_OK;
}

}  // extern "C"

// ---- facet centres of a batch of polytopes (geometric algorithm) --------------------------------------------------------------
extern "C" int mpc_facet_centres(int32_t device, int32_t n_t, int64_t n_regions, const int64_t *row_off, const double *ef_rows, double *centre,
                                 double *radius, int32_t *status) {
    if (n_t < 1 || n_regions < 0 || !row_off || !centre || !radius || !status) return fail(nullptr, MPC_ERR_INVALID, "bad argument");
    const long long rows = n_regions ? row_off[n_regions] : 0;
    if (rows == 0) return MPC_OK;
    if (!ef_rows) return fail(nullptr, MPC_ERR_INVALID, "bad argument");
    int ndev = 0;
    if ((ndev = device_count_cached()) < 1) return fail(nullptr, MPC_ERR_HIP, "no HIP device available (libmpcombi_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, MPC_ERR_INVALID, "device index out of range");
    HIP_TRY(nullptr, hipSetDevice(device));
    int m_max = 0;
    std::vector<int32_t> reg((size_t)rows);
    for (long long r = 0; r < n_regions; ++r) {
        m_max = std::max<int>(m_max, (int)(row_off[r + 1] - row_off[r]));
        for (long long i = row_off[r]; i < row_off[r + 1]; ++i) reg[(size_t)i] = (int32_t)r;
    }
    const int n = n_t + 1, ld = odd_at_least(n + 3), m = m_max + 1;
    const size_t lds = (((size_t)(m + 1) * ld * 8 + (size_t)(ld + 1 + 3 * (m + 2)) * 4) + 15) & ~size_t(15);
    if (lds > 160 * 1024) return fail(nullptr, MPC_ERR_INVALID, "a region has too many rows for the 160 KiB LDS of one CU");
    if (lds > 48 * 1024) HIP_TRY(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(k_facet_centres), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DevBuf d_ef, d_off, d_reg, d_c, d_r, d_s, d_w;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t err) { if (e == hipSuccess && err != hipSuccess) e = err; return err == hipSuccess; };
    chk(d_ef.ensure((size_t)rows * (n_t + 1) * 8, nullptr)); chk(d_off.ensure((size_t)(n_regions + 1) * 8, nullptr)); chk(d_reg.ensure((size_t)rows * 4, nullptr));
    chk(d_c.ensure((size_t)rows * n_t * 8, nullptr)); chk(d_r.ensure((size_t)rows * 8, nullptr)); chk(d_s.ensure((size_t)rows * 4, nullptr)); chk(d_w.ensure(256, nullptr));
    if (e == hipSuccess) {
        chk(hipMemcpy(d_ef.p, ef_rows, (size_t)rows * (n_t + 1) * 8, hipMemcpyHostToDevice));
        chk(hipMemcpy(d_off.p, row_off, (size_t)(n_regions + 1) * 8, hipMemcpyHostToDevice));
        chk(hipMemcpy(d_reg.p, reg.data(), (size_t)rows * 4, hipMemcpyHostToDevice));
        chk(hipMemset(d_w.p, 0, 4));
    }
    if (e == hipSuccess) {
        const int per_cu = std::max(1, std::min(32, (int)((160 * 1024) / lds)));
        const dim3 g((unsigned)std::min<long long>(rows, (long long)cu_count(device) * per_cu)), b(64);
        hipLaunchKernelGGL(k_facet_centres, g, b, lds, nullptr, rows, (int)n_t, m_max, ld, d_ef.as<double>(), d_off.as<long long>(), d_reg.as<int32_t>(),
                           d_c.as<double>(), d_r.as<double>(), d_s.as<int32_t>(), d_w.as<unsigned int>());
        chk(hipGetLastError());
        chk(hipMemcpy(centre, d_c.p, (size_t)rows * n_t * 8, hipMemcpyDeviceToHost));
        chk(hipMemcpy(radius, d_r.p, (size_t)rows * 8, hipMemcpyDeviceToHost));
        chk(hipMemcpy(status, d_s.p, (size_t)rows * 4, hipMemcpyDeviceToHost));
    }
    (void)hipDeviceSynchronize();
    for (DevBuf *bf : {&d_ef, &d_off, &d_reg, &d_c, &d_r, &d_s, &d_w}) bf->release();
    if (e != hipSuccess) return fail(nullptr, MPC_ERR_HIP, std::string("mpc_facet_centres: ") + hipGetErrorString(e));
    return MPC_OK;
}

// ---- batched LPs ------------------------------------------------------------------------------------------------
static int lp_batch_impl(int32_t device, int64_t n_lp, int32_t m, int32_t n, const double *A, int32_t shared_A, const double *b,
                         int32_t shared_b, const double *c, int32_t shared_c, const uint8_t *eq, int32_t *status, double *x,
                         double *obj, int32_t *iters, int32_t *tight) {
    if (n_lp < 0 || m < 1 || n < 1 || !A || !b || !eq || !status) return fail(nullptr, MPC_ERR_INVALID, "bad argument");
    if (n_lp == 0) return MPC_OK;
    int ndev = 0;
    if ((ndev = device_count_cached()) < 1) return fail(nullptr, MPC_ERR_HIP, "no HIP device available (libmpcombi_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, MPC_ERR_INVALID, "device index out of range");
    HIP_TRY(nullptr, hipSetDevice(device));
    const int ld = odd_at_least(n + 3);
    const size_t lds = (((size_t)(m + 1) * ld * 8 + (size_t)(ld + 1 + 3 * (m + 2)) * 4) + 15) & ~size_t(15);
    if (lds > 160 * 1024) return fail(nullptr, MPC_ERR_INVALID, "LP does not fit the 160 KiB LDS of one CU");
    if (lds > 48 * 1024) HIP_TRY(nullptr, hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int n_cu = cu_count(device);
    const size_t szA = (shared_A ? 1 : (size_t)n_lp) * m * n * 8, szb = (shared_b ? 1 : (size_t)n_lp) * m * 8;
    const size_t szc = c ? (shared_c ? 1 : (size_t)n_lp) * n * 8 : 0, szeq = (size_t)n_lp * m;
    DevBuf bA, bb, bc, beq, bst, bit, bx, bobj, bwork, btight;   // pooled blocks (see dev_pool_take)
    int rc = MPC_OK;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t err) { if (e == hipSuccess && err != hipSuccess) e = err; return err == hipSuccess; };
    chk(bA.ensure(szA, nullptr)); chk(bb.ensure(szb, nullptr));
    if (c) chk(bc.ensure(szc, nullptr));
    chk(beq.ensure(szeq, nullptr)); chk(bst.ensure((size_t)n_lp * 4, nullptr)); chk(bit.ensure((size_t)n_lp * 4, nullptr));
    chk(bx.ensure((size_t)n_lp * n * 8, nullptr)); chk(bobj.ensure((size_t)n_lp * 8, nullptr)); chk(bwork.ensure(4, nullptr));
    if (tight) chk(btight.ensure((size_t)n_lp * m * 4, nullptr));
    double *dA = bA.as<double>(), *db = bb.as<double>(), *dc = c ? bc.as<double>() : nullptr, *dx = bx.as<double>(), *dobj = bobj.as<double>();
    uint8_t *deq = beq.as<uint8_t>();
    int32_t *dst = bst.as<int32_t>(), *dit = bit.as<int32_t>(), *dtight = tight ? btight.as<int32_t>() : nullptr;
    unsigned int *dwork = bwork.as<unsigned int>();
    // own (pooled) stream: callers on other threads / handles keep running (no device-wide synchronisation)
    hipStream_t st = nullptr;
    chk(pooled_stream(&st));
    if (e == hipSuccess) {
        chk(hipMemcpyAsync(dA, A, szA, hipMemcpyHostToDevice, st)); chk(hipMemcpyAsync(db, b, szb, hipMemcpyHostToDevice, st));
        if (c) chk(hipMemcpyAsync(dc, c, szc, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(deq, eq, szeq, hipMemcpyHostToDevice, st)); chk(hipMemsetAsync(dwork, 0, 4, st));
    }
    if (e == hipSuccess) {
        const int grid = (int)std::min<long long>(n_lp, (long long)std::max(n_cu, 1) * waves_per_cu((int)lds));
        hipLaunchKernelGGL(k_lp_batch, dim3(grid), dim3(64), lds, st, (long long)n_lp, m, n, ld, dA, shared_A, db, shared_b, dc, shared_c, deq, dst, dx, dobj, dit, dtight, dwork);
        chk(hipGetLastError());
        chk(hipMemcpyAsync(status, dst, (size_t)n_lp * 4, hipMemcpyDeviceToHost, st));
        if (x) chk(hipMemcpyAsync(x, dx, (size_t)n_lp * n * 8, hipMemcpyDeviceToHost, st));
        if (obj) chk(hipMemcpyAsync(obj, dobj, (size_t)n_lp * 8, hipMemcpyDeviceToHost, st));
        if (iters) chk(hipMemcpyAsync(iters, dit, (size_t)n_lp * 4, hipMemcpyDeviceToHost, st));
        if (tight) chk(hipMemcpyAsync(tight, dtight, (size_t)n_lp * m * 4, hipMemcpyDeviceToHost, st));
    }
    if (st) { hipError_t es = hipStreamSynchronize(st); if (e == hipSuccess && es != hipSuccess) e = es; }
    if (e != hipSuccess) { rc = fail(nullptr, MPC_ERR_HIP, std::string("mpc_lp_solve_batch: ") + hipGetErrorString(e)); (void)hipDeviceSynchronize(); }
    if (st) return_stream(st);
    for (DevBuf *q : {&bA, &bb, &bc, &beq, &bst, &bit, &bx, &bobj, &bwork, &btight}) q->release();
    return rc;
}

extern "C" int mpc_lp_solve_batch(int32_t device, int64_t n_lp, int32_t m, int32_t n, const double *A, int32_t shared_A, const double *b,
                                  int32_t shared_b, const double *c, int32_t shared_c, const uint8_t *eq, int32_t *status, double *x,
                                  double *obj, int32_t *iters) {
    return lp_batch_impl(device, n_lp, m, n, A, shared_A, b, shared_b, c, shared_c, eq, status, x, obj, iters, nullptr);
}

// ---- point location ---------------------------------------------------------------------------------------------------
// ---- the QP of the program at fixed parameter points (qp.hpp) ------------------------------------------------------------------
extern "C" int mpc_qp_solve_batch(mpc_handle *h, int64_t m, const double *theta, int32_t *status, double *x, double *lambda, uint8_t *active,
                                  int32_t *iters) {
    if (!h || m < 0 || (m > 0 && (!theta || !status))) return MPC_ERR_INVALID;
    if (!h->is_qp || h->kkt_mode != 0) return fail(h, MPC_ERR_INVALID, "mpc_qp_solve_batch needs a positive definite Q");
    if (m == 0) return MPC_OK;
    { std::lock_guard<std::mutex> lk(h->wm); if (h->w_busy) return fail(h, MPC_ERR_STATE, "a level started with mpc_level_start is still running"); }
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = h->stream;
    const int nc = h->n_c, nt = h->n_t, nx = h->n_x;
    const int ld = odd_at_least(nc + 3);
    const size_t lds = ((size_t)(nc + 1) * ld + nc) * sizeof(double) + (size_t)(ld + 1 + 2 * (nc + 2) + 2) * sizeof(int32_t) + 16;
    if (lds > 160 * 1024) return fail(h, MPC_ERR_INVALID, "the QP tableau (n_c x n_c) does not fit the 160 KiB LDS of one CU");
    if (lds > 48 * 1024) HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_qp_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    DevBuf d_th, d_st, d_x, d_l, d_a, d_it;
    auto cleanup = [&]() { for (DevBuf *b : {&d_th, &d_st, &d_x, &d_l, &d_a, &d_it}) b->release(); };
    int rc = MPC_OK;
    do {
#define QP_TRY(expr) { hipError_t e__ = (expr); if (e__ != hipSuccess) { rc = fail(h, MPC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); break; } }
        QP_TRY(d_th.ensure((size_t)m * nt * sizeof(double), st));
        QP_TRY(d_st.ensure((size_t)m * sizeof(int32_t), st));
        QP_TRY(d_it.ensure((size_t)m * sizeof(int32_t), st));
        if (x) QP_TRY(d_x.ensure((size_t)m * nx * sizeof(double), st));
        if (lambda) QP_TRY(d_l.ensure((size_t)m * nc * sizeof(double), st));
        if (active) QP_TRY(d_a.ensure((size_t)m * nc, st));
        QP_TRY(hipMemcpyAsync(d_th.p, theta, (size_t)m * nt * sizeof(double), hipMemcpyHostToDevice, st));
        QP_TRY(hipMemsetAsync(h->scratch.p, 0, sizeof(unsigned int), st));
        const int per_cu = std::max(1, std::min(16, (int)((160 * 1024) / lds)));
        const dim3 g((unsigned)std::min<long long>(m, (long long)h->n_cu * per_cu)), b(64);
        hipLaunchKernelGGL(k_qp_batch, g, b, lds, st, (long long)m, nc, h->n_eq, nt, nx, ld, h->Pv.W, h->Pv.UV, h->Pv.X0H, h->Pv.Gt, d_th.as<double>(),
                           d_st.as<int32_t>(), x ? d_x.as<double>() : (double *)nullptr, lambda ? d_l.as<double>() : (double *)nullptr,
                           active ? d_a.as<uint8_t>() : (uint8_t *)nullptr, d_it.as<int32_t>(), h->scratch.as<unsigned int>());
        QP_TRY(hipGetLastError());
        QP_TRY(hipMemcpyAsync(status, d_st.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (iters) QP_TRY(hipMemcpyAsync(iters, d_it.p, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (x) QP_TRY(hipMemcpyAsync(x, d_x.p, (size_t)m * nx * sizeof(double), hipMemcpyDeviceToHost, st));
        if (lambda) QP_TRY(hipMemcpyAsync(lambda, d_l.p, (size_t)m * nc * sizeof(double), hipMemcpyDeviceToHost, st));
        if (active) QP_TRY(hipMemcpyAsync(active, d_a.p, (size_t)m * nc, hipMemcpyDeviceToHost, st));
        QP_TRY(hipStreamSynchronize(st));
#undef QP_TRY
    } while (0);
    if (rc != MPC_OK) (void)hipStreamSynchronize(st);
    cleanup();
    return rc;
}

struct mpc_locator {
    int device = 0, n_x = 0, n_t = 0;
    long long n_regions = 0, n_rows = 0;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    DevBuf row_off, row_region, row_end, ef, xlaw, Q, c, H, theta, region, x;
    // adjacency for the walk (mpc_locator_set_adjacency): active-set masks in region order and sorted, facet kind / id per row
    DevBuf masks, sorted_masks, sorted_region, row_info, theta2, region2;
    int mask_words = 0, n_c = 0;
    bool has_adj = false;
    long long last_unresolved = 0;   // points of the last walk query that went to the list scan
    bool hasQ = false, hasc = false, hasH = false;
};

static int locator_fill(mpc_locator *L, int64_t n_regions, const int64_t *row_off, const double *ef_rows, const double *xlaw, const double *Q,
                        const double *c, const double *H);
extern "C" int mpc_locator_destroy(mpc_locator *L);

extern "C" int mpc_locator_create(int32_t device, int32_t n_x, int32_t n_t, int64_t n_regions, const int64_t *row_off, const double *ef_rows,
                                  const double *xlaw, const double *Q, const double *c, const double *H, mpc_locator **out) {
    if (!out || n_x < 1 || n_t < 1 || n_t > 16 || n_regions < 0 || (n_regions > 0 && (!row_off || !ef_rows || !xlaw)))
        return fail(nullptr, MPC_ERR_INVALID, "mpc_locator_create: bad arguments (1 <= n_t <= 16)");
    HIP_TRY(nullptr, hipSetDevice(device));
    mpc_locator *L = new mpc_locator();
    L->device = device; L->n_x = n_x; L->n_t = n_t; L->n_regions = n_regions;
    const int rc_fill = locator_fill(L, n_regions, row_off, ef_rows, xlaw, Q, c, H);
    if (rc_fill != MPC_OK) { (void)mpc_locator_destroy(L); return rc_fill; }   // one cleanup path: nothing leaks on failure
    *out = L;
    return MPC_OK;
}

static int locator_fill(mpc_locator *L, int64_t n_regions, const int64_t *row_off, const double *ef_rows, const double *xlaw, const double *Q,
                        const double *c, const double *H) {
    const int n_x = L->n_x, n_t = L->n_t;
    HIP_TRY(nullptr, hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking));
    HIP_TRY(nullptr, hipEventCreate(&L->e0));
    HIP_TRY(nullptr, hipEventCreate(&L->e1));
    const long long rows = n_regions ? row_off[n_regions] : 0;
    L->n_rows = rows;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(std::max<size_t>(bytes, 8), L->stream);
        if (e != hipSuccess || !bytes) return e;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, L->stream);
    };
    const long long zero = 0;
    HIP_TRY(nullptr, up(L->row_off, n_regions ? (const void *)row_off : (const void *)&zero, (size_t)(n_regions + 1) * sizeof(int64_t)));
    HIP_TRY(nullptr, up(L->ef, ef_rows, (size_t)rows * (n_t + 1) * sizeof(double)));
    std::vector<int32_t> rr((size_t)std::max<long long>(rows, 1)), re((size_t)std::max<long long>(rows, 1));
    for (long long r = 0; r < n_regions; ++r)
        for (long long i = row_off[r]; i < row_off[r + 1]; ++i) { rr[(size_t)i] = (int32_t)r; re[(size_t)i] = (int32_t)row_off[r + 1]; }
    HIP_TRY(nullptr, up(L->row_region, rr.data(), (size_t)rows * sizeof(int32_t)));
    HIP_TRY(nullptr, up(L->row_end, re.data(), (size_t)rows * sizeof(int32_t)));
    HIP_TRY(nullptr, up(L->xlaw, xlaw, (size_t)n_regions * n_x * (n_t + 1) * sizeof(double)));
    if (Q) { HIP_TRY(nullptr, up(L->Q, Q, (size_t)n_x * n_x * sizeof(double))); L->hasQ = true; }
    if (c) { HIP_TRY(nullptr, up(L->c, c, (size_t)n_x * sizeof(double))); L->hasc = true; }
    if (H) { HIP_TRY(nullptr, up(L->H, H, (size_t)n_x * n_t * sizeof(double))); L->hasH = true; }
    HIP_TRY(nullptr, hipStreamSynchronize(L->stream));
    return MPC_OK;
}

extern "C" int mpc_locator_set_adjacency(mpc_locator *L, int32_t mask_words, int32_t n_c, const uint64_t *masks, const int32_t *row_info) {
    if (!L || !masks || !row_info || (mask_words != 2 && mask_words != 4) || n_c < 1 || n_c > 64 * mask_words) return MPC_ERR_INVALID;
    HIP_TRY(nullptr, hipSetDevice(L->device));
    const long long n = L->n_regions, rows = L->n_rows;
    if (n <= 0) return MPC_OK;
    const int mw = mask_words;
    // the mask table sorted ascending (most significant word last), with the region each mask belongs to
    std::vector<int32_t> order((size_t)n);
    for (long long i = 0; i < n; ++i) order[(size_t)i] = (int32_t)i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        for (int j = mw - 1; j >= 0; --j) { const uint64_t va = masks[(size_t)a * mw + j], vb = masks[(size_t)b * mw + j]; if (va != vb) return va < vb; }
        return a < b;
    });
    std::vector<uint64_t> sorted((size_t)n * mw);
    for (long long i = 0; i < n; ++i) for (int j = 0; j < mw; ++j) sorted[(size_t)i * mw + j] = masks[(size_t)order[(size_t)i] * mw + j];
    for (long long i = 1; i < n; ++i) {   // two regions with one active set: no unique neighbour, no walk
        bool same = true;
        for (int j = 0; j < mw; ++j) same = same && sorted[(size_t)i * mw + j] == sorted[(size_t)(i - 1) * mw + j];
        if (same) return fail(nullptr, MPC_ERR_INVALID, "mpc_locator_set_adjacency: two regions share one active set");
    }
    hipStream_t st = L->stream;
    auto up = [&](DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(std::max<size_t>(bytes, 8), st);
        if (e != hipSuccess || !bytes) return e;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st);
    };
    HIP_TRY(nullptr, up(L->masks, masks, (size_t)n * mw * sizeof(uint64_t)));
    HIP_TRY(nullptr, up(L->sorted_masks, sorted.data(), (size_t)n * mw * sizeof(uint64_t)));
    HIP_TRY(nullptr, up(L->sorted_region, order.data(), (size_t)n * sizeof(int32_t)));
    HIP_TRY(nullptr, up(L->row_info, row_info, (size_t)rows * sizeof(int32_t)));
    HIP_TRY(nullptr, hipStreamSynchronize(st));
    L->mask_words = mw;
    L->n_c = n_c;
    L->has_adj = true;
    return MPC_OK;
}

extern "C" int mpc_locator_query(mpc_locator *L, int64_t m, const double *theta, double tol, int32_t flags, int64_t *region, double *x,
                                 float *ms_locate) {
    if (!L || m < 0 || (m > 0 && (!theta || !region))) return MPC_ERR_INVALID;
    if (ms_locate) *ms_locate = 0.0f;
    if (m == 0) return MPC_OK;
    HIP_TRY(nullptr, hipSetDevice(L->device));
    hipStream_t st = L->stream;
    const int nt = L->n_t, nx = L->n_x;
    HIP_TRY(nullptr, L->theta.ensure((size_t)m * nt * sizeof(double), st));
    HIP_TRY(nullptr, L->region.ensure((size_t)m * sizeof(long long), st));
    HIP_TRY(nullptr, hipMemcpyAsync(L->theta.p, theta, (size_t)m * nt * sizeof(double), hipMemcpyHostToDevice, st));
    const dim3 g((unsigned)((m + 255) / 256)), b(256);
    const double *Q = L->hasQ ? L->Q.as<double>() : nullptr, *c = L->hasc ? L->c.as<double>() : nullptr, *H = L->hasH ? L->H.as<double>() : nullptr;
    HIP_TRY(nullptr, hipEventRecord(L->e0, st));
#define MPC_LOCATE(NT_, M_, TH_, OUT_) hipLaunchKernelGGL((k_locate<NT_>), dim3((unsigned)(((M_) + 255) / 256)), b, 0, st, (long long)(M_), nt, nx, L->n_regions, L->n_rows, L->row_region.as<int32_t>(), \
                                           L->row_end.as<int32_t>(), L->ef.as<double>(), L->xlaw.as<double>(), Q, c, H, TH_, tol, \
                                           (int)(flags & MPC_LOCATE_OVERLAPPING), (int)((flags & MPC_LOCATE_INCLUSIVE) != 0), OUT_)
#define MPC_LOCATE_ANY(M_, TH_, OUT_) do { if (nt <= 4) MPC_LOCATE(4, M_, TH_, OUT_); else if (nt <= 8) MPC_LOCATE(8, M_, TH_, OUT_); else MPC_LOCATE(16, M_, TH_, OUT_); } while (0)
    const bool walk = (flags & MPC_LOCATE_WALK) && L->has_adj && !(flags & (MPC_LOCATE_OVERLAPPING | MPC_LOCATE_INCLUSIVE)) && L->n_regions > 0 && nt <= 16;
    if (!walk) {
        MPC_LOCATE_ANY(m, L->theta.as<double>(), L->region.as<long long>());
        HIP_TRY(nullptr, hipGetLastError());
        HIP_TRY(nullptr, hipEventRecord(L->e1, st));
        HIP_TRY(nullptr, hipMemcpyAsync(region, L->region.p, (size_t)m * sizeof(long long), hipMemcpyDeviceToHost, st));
    } else {
        // walk through adjacent regions; what the walk cannot resolve goes to the list scan
        const int max_steps = 384;   // walks are tens of steps long; what is still open then (points outside the solution, mostly) goes to k_locate_few
#define MPC_WALK(NT_, MW_) hipLaunchKernelGGL((k_locate_walk<NT_, MW_>), g, b, 0, st, (long long)m, nt, L->n_regions, L->row_off.as<long long>(), L->ef.as<double>(), \
                                              L->row_info.as<int32_t>(), L->masks.as<unsigned long long>(), L->sorted_masks.as<unsigned long long>(), \
                                              L->sorted_region.as<int32_t>(), L->theta.as<double>(), tol, 0, max_steps, L->n_c, L->region.as<long long>())
        if (L->mask_words == 2) { if (nt <= 4) MPC_WALK(4, 2); else if (nt <= 8) MPC_WALK(8, 2); else MPC_WALK(16, 2); }
        else { if (nt <= 4) MPC_WALK(4, 4); else if (nt <= 8) MPC_WALK(8, 4); else MPC_WALK(16, 4); }
#undef MPC_WALK
        HIP_TRY(nullptr, hipGetLastError());
        HIP_TRY(nullptr, hipMemcpyAsync(region, L->region.p, (size_t)m * sizeof(long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(nullptr, hipStreamSynchronize(st));
        std::vector<long long> open;
        for (long long p = 0; p < m; ++p) if (region[p] == -2) open.push_back(p);
        if (!open.empty()) {
            const long long mo = (long long)open.size();
            std::vector<double> tho((size_t)mo * nt);
            for (long long i = 0; i < mo; ++i) std::memcpy(&tho[(size_t)i * nt], theta + (size_t)open[(size_t)i] * nt, sizeof(double) * nt);
            std::vector<long long> ro((size_t)mo);
            HIP_TRY(nullptr, L->theta2.ensure((size_t)mo * nt * sizeof(double), st));
            HIP_TRY(nullptr, L->region2.ensure((size_t)mo * sizeof(long long), st));
            HIP_TRY(nullptr, hipMemcpyAsync(L->theta2.p, tho.data(), (size_t)mo * nt * sizeof(double), hipMemcpyHostToDevice, st));
            if (mo <= 16384) {
                // few points: every (point, region) pair in parallel, first containing region by atomicMin
                HIP_TRY(nullptr, hipMemsetAsync(L->region2.p, 0xff, (size_t)mo * sizeof(long long), st));   // = "none yet" (max u64)
                const dim3 gf((unsigned)((L->n_regions + 255) / 256), (unsigned)mo);
                if (nt <= 4) hipLaunchKernelGGL((k_locate_few<4>), gf, b, 0, st, mo, nt, L->n_regions, L->row_off.as<long long>(), L->ef.as<double>(), L->theta2.as<double>(), tol, L->region2.as<long long>());
                else if (nt <= 8) hipLaunchKernelGGL((k_locate_few<8>), gf, b, 0, st, mo, nt, L->n_regions, L->row_off.as<long long>(), L->ef.as<double>(), L->theta2.as<double>(), tol, L->region2.as<long long>());
                else hipLaunchKernelGGL((k_locate_few<16>), gf, b, 0, st, mo, nt, L->n_regions, L->row_off.as<long long>(), L->ef.as<double>(), L->theta2.as<double>(), tol, L->region2.as<long long>());
            } else {
                MPC_LOCATE_ANY(mo, L->theta2.as<double>(), L->region2.as<long long>());
            }
            HIP_TRY(nullptr, hipGetLastError());
            HIP_TRY(nullptr, hipMemcpyAsync(ro.data(), L->region2.p, (size_t)mo * sizeof(long long), hipMemcpyDeviceToHost, st));
            HIP_TRY(nullptr, hipStreamSynchronize(st));
            for (long long i = 0; i < mo; ++i) region[open[(size_t)i]] = ro[(size_t)i];   // -1 (all ones) where no region contains the point
            HIP_TRY(nullptr, hipMemcpyAsync(L->region.p, region, (size_t)m * sizeof(long long), hipMemcpyHostToDevice, st));   // k_evaluate reads it
        }
        L->last_unresolved = (long long)open.size();
        HIP_TRY(nullptr, hipEventRecord(L->e1, st));
    }
#undef MPC_LOCATE_ANY
#undef MPC_LOCATE
    if (x) {
        HIP_TRY(nullptr, L->x.ensure((size_t)m * nx * sizeof(double), st));
        const long long tot = (long long)m * nx;
        hipLaunchKernelGGL(k_evaluate, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (long long)m, nt, nx, L->xlaw.as<double>(), L->theta.as<double>(),
                           L->region.as<long long>(), L->x.as<double>());
        HIP_TRY(nullptr, hipGetLastError());
        HIP_TRY(nullptr, hipMemcpyAsync(x, L->x.p, (size_t)m * nx * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(nullptr, hipStreamSynchronize(st));
    if (ms_locate) HIP_TRY(nullptr, hipEventElapsedTime(ms_locate, L->e0, L->e1));
    return MPC_OK;
}

extern "C" int mpc_locator_destroy(mpc_locator *L) {
    if (!L) return MPC_OK;
    (void)hipSetDevice(L->device);
    if (L->stream) (void)hipStreamSynchronize(L->stream);
    for (DevBuf *b : {&L->row_off, &L->row_region, &L->row_end, &L->ef, &L->xlaw, &L->Q, &L->c, &L->H, &L->theta, &L->region, &L->x, &L->masks, &L->sorted_masks,
                      &L->sorted_region, &L->row_info, &L->theta2, &L->region2}) b->release();
    if (L->e0) (void)hipEventDestroy(L->e0);
    if (L->e1) (void)hipEventDestroy(L->e1);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    delete L;
    return MPC_OK;
}
