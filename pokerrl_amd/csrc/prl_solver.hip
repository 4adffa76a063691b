// prl_solver_*: device-resident tabular CFR / best-response solver behind the C ABI (include/pokerrl_hip.h section 5).
// Host orchestration only: uploads the flat tree, owns the HIP stream and the HBM arrays, issues the kernels in the
// reference's order (_CFRBase.py:110-134). Nothing here computes on the CPU.
//
// Two engines share this front end:
//   LEVELS  every per-node vector in HBM, one kernel per tree level (prl_tree_kernels.hip). Any supported tree; exposes
//           node.reach_probs / node.ev / node.ev_br / best-response indices.
//   FUSED   Flop5Holdem-shaped trees with CFR+: the pre-deal trunk (a handful of nodes) runs on the LEVELS kernels with
//           the chance node as a leaf; each board subtree is walked on chip by one workgroup (prl_fhp_kernels.hip); the
//           boards' root values are summed in the canonical nested order into the chance node.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#if !defined(PRL_EMU)
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is bound at run time (prl_rccl_api), so that single-GPU users never load it
#endif

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "prl_cards.h"
#include "prl_device.h"
#include "prl_fhp.h"
#include "prl_host.h"
#include "prl_kernels.h"
#include "prl_policy.h"
#include "prl_rt.h"
#include "prl_solver_types.h"
#include "prl_st.h"

#define PRL_NODE_LEAF 4  // trunk view of the chance node in the FUSED engine: values are written by the chance sum

struct prl_solver {
    PrlFlatTree ft;           // host copy of the tree the device kernels walk (the trunk only, for the FUSED engine)
    PrlDevTree T{};
    PrlDevState S{};
    PrlDevState Seval{};      // scratch per-node vectors for the average-strategy evaluation (allocated lazily)
    bool eval_ready = false;
    hipStream_t stream = nullptr;
    std::vector<void*> allocs;
    std::vector<void*> vmm;  // PRL_VMM_SHUFFLE_MB: shuffled virtual-memory-management ranges (PrlVmmRange*, dev_alloc)
    int32_t* d_term_nodes = nullptr;
    int n_term = 0;
    int32_t* d_nodes_p[2] = {nullptr, nullptr};
    int n_nodes_p[2] = {0, 0};
    int32_t* d_col_node = nullptr;
    int variant = PRL_CFR_PLUS, delay = 0, iter = 0;
    bool ev_valid = false;    // S.ev / S.ev_br / S.expl correspond to the current strategy + reach
    bool root_reach_set[2] = {false, false};  // the root's reach (a constant) is in place: S / any other state
    float* expl_copy_dst = nullptr;  // where the next evaluation's exploitability goes besides S.expl (its slot of the history)
    float* d_expl_hist = nullptr;  // [cap][2] current-strategy exploitability after every iteration
    int hist_cap = 0;
    size_t bytes_allocated = 0;
    // full-tree sizes (what the caller sees)
    int full_nodes = 0, full_cols = 0, R = 0;
    // ---- FUSED engine ----
    bool fused = false;
    // sharded solve (prl_solver_create_sharded)
    int world = 1, rank = 0, xlevel = 0, n_units = 0;  // summation level exchanged, units per rank (the exchange's fixed stride)
    int n_units_all = 0;                               // units over all ranks: (world - 1) * n_units + the last rank's (ragged shards)
    uint64_t fingerprint = 0;  // boards + game + rules + (world, rank): what a checkpoint must match besides the array shapes
    prl_exchange_fn exchange = nullptr;
    bool exchange_async = false;  // the callback enqueues on s->stream: no host synchronisation around it
    bool block_sum = true;        // the board pass sums its root vectors per 32-board block (PRL_FHP_NO_BLOCK_SUM: per-board rows, tests)
    void* exchange_user = nullptr;
    float *d_xlocal = nullptr, *d_xgather = nullptr;
    bool have_half = false;      // FUSED steady state: seat 1's half of the exploitability of the current iterate is in d_half
    float* d_board_out = nullptr;  // [n_boards][<= 4][R] root vectors of the last board pass
    float* d_row_sum = nullptr;    // [<= 4][R] their canonical sum
    // weighted boards / suit isomorphism (prl_solver_create_weighted)
    const float* d_board_w = nullptr;  // [n_boards] chance_prob * multiplicity
    bool symmetrize = false;
    float* d_row_sym = nullptr;        // [<= 4][R] the symmetrised sum
    const int32_t *d_sym_class_of = nullptr, *d_sym_class_start = nullptr, *d_sym_class_hands = nullptr;
    float* d_half = nullptr;     // [R] chance-summed seat-1 value under its new strategy, [R] its best response
    // LEVELS engine: one captured hipGraph of a whole iteration, replayed per iteration (launch-bound small trees)
    PrlIterDev* d_ip = nullptr;
    const int32_t* d_level_start = nullptr;  // small-tree path: BFS level offsets on the device
    bool small_tree = false;                 // 1-hole-card tree small enough for the single-workgroup iteration kernel
    void* levels_graph_exec = nullptr;  // hipGraphExec_t
    bool graphs_off = false;
    int avg_pending[2] = {-1, -1};  // FUSED Vanilla / Linear: iteration whose average update of that seat still has to run
    bool time_passes = false;   // prl_solver_time_iterations: bracket every board-pass launch with events
    std::vector<hipEvent_t> pass_events;
    bool expl_pending = false;  // FUSED, inside prl_solver_iterations: exploitability of the current iterate not evaluated yet
    PrlFhpParams fp{};
    int chance_trunk = -1;    // trunk id of the chance node
    float* d_sum_scratch = nullptr;
    double* d_user_strategy = nullptr;  // [full_cols][R], explicit strategy (prl_solver_set_strategy), lazily allocated
    float* d_user_strategy32 = nullptr; // FUSED: an explicit float32 strategy stays float32, [full_cols][R] like the regrets (best-response pass)
    int user_strategy_f64 = -1;         // -1: strategy comes from regrets / uniform
    int src[2] = {PRL_SRC_UNIFORM64, PRL_SRC_UNIFORM64};
    bool board_avg_f64 = false;
    bool board_avg_stale = false;  // FUSED Vanilla / Linear: avg_sum moved on, the avg columns of the boards have not been recomputed yet
    float* d_regret = nullptr;  // [full_cols][R]
    double* d_avg = nullptr;    // [full_cols][R] (avg_f32: the trunk's columns only)
    float* d_avg32 = nullptr;   // opt-in (PRL_SOLVER_AVG_F32): the board columns' running average stored as float32, [full_cols][R]
    bool avg_f32 = false;
    long long n_exchanges = 0;  // all-gathers done so far (PRL_SF_EXCHANGES)
    void* rccl_comm = nullptr;  // sharded solve with the library's own exchange (prl_solver_create_sharded_rccl): ncclComm_t
    // ---- per-street fused engine (prl_st.h): `fused` with the board pass replaced by a sweep over the streets ----
    bool streets = false;
    PrlStPlanHost st;
    PrlStParams sp{};                  // what every street launch shares (plans, sizes)
    struct StLevelDev { PrlStInst* inst = nullptr; } st_dev[PRL_ST_MAX_GROUPS];                                    // per (street, shape) group
    struct StStreetDev { float* leaf_reach = nullptr; float* val = nullptr; } st_str[PRL_ST_MAX_LEVELS + 1];       // per street: the groups share them
    // run-out chains below all-in calls (prl_st.h "MIXED STREETS"): a decision-free forest on the LEVELS kernels
    PrlFlatTree chain_ft;
    PrlDevTree Tc{};
    float* d_chain_ev = nullptr;       // [forest nodes][2][R]: values of the forest's inner nodes (its roots' values go straight to their streets' rows)
    PrlStChainTerm* d_chain_terms = nullptr;
    PrlStChainBundle* d_chain_bundles = nullptr;
    hipStream_t chain_stream = nullptr;      // the run-out forest is evaluated beside the last street's passes (vector issue beside HBM streaming)
    hipEvent_t chain_fork = nullptr, chain_join = nullptr;
    std::vector<int32_t> chain_level_start;  // the forest's chance nodes by level (Tc.level_nodes lists them, not the showdowns)
    int n_chain_term = 0, n_chain_bundles = 0, n_chain = 0;
    PrlStChainDev chain_dev{};
    int32_t* d_trunk_leaves = nullptr; // trunk ids of the trunk's chance leaves
    int n_trunk_leaves = 1;
    std::vector<int32_t> col_dfs;      // internal column -> flat-tree (DFS) column; empty = identity (every other engine)
    std::vector<int32_t> col_int;      // its inverse, built by the first prl_solver_get_cols
    // ---- single-deal fused engine: SORTED STORAGE of the board columns (prl_fhp.h) ----
    // every column array is [trunk columns][R] in hand order, then -- from element `board_ofs` -- the board region
    // [n_boards][ncb][PRL_FHP_NP] in each board's rank-sorted order, live hands only: `col_elems` elements in all
    bool sorted = false;
    int col_base = 0, ncb = 0;         // global column id of board 0's first column, columns per board
    size_t board_ofs = 0, col_elems = 0;
    double blk_avg[4] = {0., 0., 0., 0.};  // CFR+ running average of a hand its board blocks, by action count: the board pass keeps no
                                       // storage for it (regrets 0 for ever -> uniform strategy -> a scalar recurrence per action count)
    float* d_user_blocked32 = nullptr; // what set_strategy's caller had for the blocked hands, [n_boards * ncb][PRL_FHP_NBLOCKED]: get returns it
    double* d_user_blocked64 = nullptr;
    char* d_stage = nullptr;           // translation buffer of get / set: `stage_boards` boards x ncb x R x 8 bytes
    int stage_boards = 0;
    double* d_fill = nullptr;          // [PRL_FHP_MAX_NODES * 3] per-column fill values of an expansion
};

namespace {

#if !defined(PRL_EMU)
// PRL_VMM_SHUFFLE_MB=c (default: 2 for sharded solves, 0 = off otherwise): a large array as ONE virtual range backed by c MB physical chunks mapped in a SHUFFLED
// order (HIP virtual memory management). The board pass streams within 10 % of what the part sustains and its speed depends on where
// its 66 GB land physically: plain hipMalloc objects of one process differ by up to 15 %, physically contiguous backing
// (hipDeviceMallocContiguous) is always the slow case. Shuffled 2 MB chunks remove the object-to-object spread on a box (12 of 12
// objects within 0.2 %) -- but at a level that is itself box-dependent: equal to the best plain objects on one box (26.1 ms), 7 %
// behind them on another (27.6 vs 25.8 ms), so it is the default only where the slowest of several objects sets the pace (sharded
// solves; profiles/r02_experiments.txt). Returns nullptr when anything fails (the caller falls back to hipMalloc).
struct PrlVmmRange { void* va; size_t size, chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
static void* vmm_alloc_shuffled(size_t bytes, size_t chunk, PrlVmmRange* out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return nullptr;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk, size = n * chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, size, chunk, nullptr, 0) != hipSuccess) return nullptr;
    out->va = va; out->size = size; out->chunk = chunk;
    std::vector<size_t> slot(n);
    for (size_t i = 0; i < n; ++i) slot[i] = i;
    unsigned long long x = 0x9E3779B97F4A7C15ull;  // fixed seed: the same mapping every run
    for (size_t i = n; i > 1; --i) { x = x * 6364136223846793005ull + 1442695040888963407ull; std::swap(slot[i - 1], slot[(size_t)((x >> 33) % i)]); }
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
        out->handles.push_back(h);
        if (hipMemMap((char*)va + slot[i] * chunk, chunk, 0, h, 0) != hipSuccess) ok = false;
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (ok && hipMemSetAccess(va, size, &acc, 1) != hipSuccess) ok = false;
    if (!ok) {
        (void)hipGetLastError();
        (void)hipMemUnmap(va, size);
        for (auto h : out->handles) (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, size);
        out->handles.clear();
        return nullptr;
    }
    return va;
}
static void vmm_free(PrlVmmRange& r) {
    (void)hipMemUnmap(r.va, r.size);
    for (auto h : r.handles) (void)hipMemRelease(h);
    (void)hipMemAddressFree(r.va, r.size);
}
#endif

#if !defined(PRL_EMU)
// RCCL bound at run time. A process that has PyTorch-ROCm loaded already has an RCCL (and ONE HIP runtime, see pokerrl_amd/_native.py):
// that one is taken; otherwise the ROCm installation's.
struct PrlRcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why, path;  // why it is not usable / the file the symbols come from
};
static const PrlRcclApi& prl_rccl_api() {
    static PrlRcclApi api = [] {
        PrlRcclApi a;
        // PRL_RCCL_LIB=<path>: bind exactly that library (a ROCm installed elsewhere, or to force the one a framework ships). Otherwise the
        // RCCL ALREADY in the process (PyTorch-ROCm's: one RCCL, one HIP runtime per process), then the loader's, then ROCm's default place.
        void* h = nullptr;
        const char* forced = getenv("PRL_RCCL_LIB");
        if (forced && *forced) {
            h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
            if (!h) {
                const char* de = dlerror();  // ONE call: dlerror() clears its state when read
                a.why = std::string("PRL_RCCL_LIB=") + forced + ": " + (de ? de : "dlopen failed");
                return a;
            }
        }
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { a.why = "librccl.so / librccl.so.1 not found (already loaded, on the loader's path, or /opt/rocm/lib); set PRL_RCCL_LIB=<path to librccl.so>"; return a; }
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
        a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy && a.GetErrorString;
        if (!a.ok) a.why = "the RCCL library bound lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy / ncclGetErrorString";
        {   // which file it is: a second RCCL beside the framework's is the classic way to a hang at the first collective
            Dl_info di;
            if (a.GetUniqueId && dladdr((void*)a.GetUniqueId, &di) && di.dli_fname) a.path = di.dli_fname;
        }
        return a;
    }();
    return api;
}
// the exchange of a solver created with prl_solver_create_sharded_rccl: one ncclAllGather on the solver's own stream -- stream-ordered
// after the partial sums, before the finishing sum; no host synchronisation, no callback into the host language
static int32_t prl_rccl_exchange(void* user, const void* local_dev, void* gathered_dev, uint64_t bytes_per_rank) {
    prl_solver* s = (prl_solver*)user;
    const ncclResult_t r = prl_rccl_api().AllGather(local_dev, gathered_dev, (size_t)bytes_per_rank, ncclInt8, (ncclComm_t)s->rccl_comm, s->stream);
    if (r != ncclSuccess) { prl_set_error(std::string("ncclAllGather: ") + prl_rccl_api().GetErrorString(r)); return 1; }
    return 0;
}
#endif

template <class T>
int dev_alloc(prl_solver* s, T** p, size_t count, bool plain = false) {  // plain: buffers handed to the exchange callback (RCCL)
    void* q = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
#if !defined(PRL_EMU)
    // default: on (2 MB chunks) for sharded solves -- a multi-GPU step lasts as long as its SLOWEST rank, and with plain allocations
    // one rank in three is a slow one -- off for a single GPU, where a well-placed plain allocation is the fastest case
    static const long vmm_env = getenv("PRL_VMM_SHUFFLE_MB") ? atol(getenv("PRL_VMM_SHUFFLE_MB")) : -1;
    const long vmm_mb = vmm_env >= 0 ? vmm_env : (s->world > 1 ? 2 : 0);
    if (!plain && vmm_mb > 0 && bytes >= ((size_t)32 << 20)) {
        PrlVmmRange* r = new PrlVmmRange();
        q = vmm_alloc_shuffled(bytes + 4096, (size_t)vmm_mb << 20, r);
        if (q) {
            s->vmm.push_back(r);
            s->bytes_allocated += bytes;
            *p = (T*)q;
            return PRL_OK;
        }
        delete r;  // (no message: plain hipMalloc is a correct fallback, only the run-to-run spread comes back)
    }
#endif
    // spare bytes at the far end: the fused engine's LDS-DMA prefetch fetches whole rows of 64 x 16 bytes, so the last board's
    // last row reads up to ~1 KB past its block
    hipError_t e = hipMalloc(&q, bytes + 4096);
    if (e != hipSuccess) {
        prl_set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        return PRL_ERR_OOM;
    }
    s->allocs.push_back(q);
    s->bytes_allocated += bytes;
    *p = (T*)q;
    return PRL_OK;
}

template <class T>
int dev_upload(prl_solver* s, const T** p, const std::vector<T>& v) {
    T* q = nullptr;
    int e = dev_alloc(s, &q, v.size());
    if (e) return e;
    if (!v.empty()) PRL_HIP_TRY(hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = q;
    return PRL_OK;
}

#define TRY(x) do { int e_ = (x); if (e_) return e_; } while (0)

// elements of one column array (regret, avg, ...): hand-order [full_cols][R], or trunk columns + the sorted board region
static size_t col_array_elems(const prl_solver* s) { return s->sorted ? s->col_elems : (size_t)s->full_cols * s->R; }

// ---- sorted storage <-> the caller's hand-order columns (single-deal fused engine), a chunk of boards at a time -----------------
static int stage_ready(prl_solver* s) {
    if (s->d_stage) return PRL_OK;
    const size_t per_board = (size_t)s->ncb * s->R * 8;
    size_t nb = ((size_t)192 << 20) / per_board;
    if (nb < 1) nb = 1;
    if (nb > (size_t)s->fp.n_boards) nb = (size_t)s->fp.n_boards;
    TRY(dev_alloc(s, &s->d_stage, nb * per_board, true));
    TRY(dev_alloc(s, &s->d_fill, (size_t)PRL_FHP_MAX_NODES * 3, true));
    s->stage_boards = (int)nb;
    return PRL_OK;
}
// local column j of a board -> the action count of its decision node
static int board_col_actions(const prl_solver* s, int j) {
    for (int d = 0; d < s->fp.n_dec; ++d)
        if (j >= s->fp.dec_col0[d] && j < s->fp.dec_col0[d] + s->fp.dec_nch[d]) return s->fp.dec_nch[d];
    return 1;
}
// boards [b0, b0 + nb) of a board region -> host columns [nb * ncb][R] (hand order), `out_elem` bytes per element (8 with elem 4: float32
// storage widened exactly). fill_by_col: value of the hands a board blocks, by local column (nullptr: 0); blocked_src: or the side array
static int sorted_get_boards(prl_solver* s, const void* region, int elem, const double* fill_by_col, const void* blocked_src, void* out, int out_elem,
                             int b0, int nb) {
    TRY(stage_ready(s));
    const void* d_fill = nullptr;
    if (fill_by_col) {
        char tmp[PRL_FHP_MAX_NODES * 3 * 8];
        for (int j = 0; j < s->ncb; ++j) {
            if (elem == 4) ((float*)tmp)[j] = (float)fill_by_col[j];
            else ((double*)tmp)[j] = fill_by_col[j];
        }
        PRL_HIP_TRY(hipMemcpyAsync(s->d_fill, tmp, (size_t)s->ncb * elem, hipMemcpyHostToDevice, s->stream));
        PRL_HIP_TRY(hipStreamSynchronize(s->stream));
        d_fill = s->d_fill;
    }
    const size_t per_board = (size_t)s->ncb * s->R;
    std::vector<float> widen;
    for (int at = 0; at < nb; at += s->stage_boards) {
        const int n = nb - at < s->stage_boards ? nb - at : s->stage_boards;
        prl_launch_fhp_expand(s->fp, region, elem, b0 + at, n, d_fill, blocked_src, s->d_stage, s->stream);
        PRL_HIP_TRY(hipGetLastError());
        char* o = (char*)out + (size_t)at * per_board * out_elem;
        if (out_elem == elem) PRL_HIP_TRY(hipMemcpyAsync(o, s->d_stage, (size_t)n * per_board * elem, hipMemcpyDeviceToHost, s->stream));
        else {
            widen.resize((size_t)n * per_board);
            PRL_HIP_TRY(hipMemcpyAsync(widen.data(), s->d_stage, (size_t)n * per_board * 4, hipMemcpyDeviceToHost, s->stream));
            PRL_HIP_TRY(hipStreamSynchronize(s->stream));
            double* od = (double*)o;
            for (size_t i = 0; i < (size_t)n * per_board; ++i) od[i] = (double)widen[i];
        }
        PRL_HIP_TRY(hipStreamSynchronize(s->stream));  // the staging buffer is reused
    }
    return PRL_OK;
}
// host columns [n_boards * ncb][R] (hand order) -> a board region (+ the caller's values for the blocked hands)
static int sorted_set_boards(prl_solver* s, const void* host_cols, int elem, void* region, void* blocked_dst) {
    TRY(stage_ready(s));
    const size_t per_board = (size_t)s->ncb * s->R;
    const int nb = s->fp.n_boards;
    for (int at = 0; at < nb; at += s->stage_boards) {
        const int n = nb - at < s->stage_boards ? nb - at : s->stage_boards;
        PRL_HIP_TRY(hipMemcpyAsync(s->d_stage, (const char*)host_cols + (size_t)at * per_board * elem, (size_t)n * per_board * elem, hipMemcpyHostToDevice, s->stream));
        // (the kernel indexes the staging buffer from its start: src row = local board)
        PrlFhpParams q = s->fp;
        prl_launch_fhp_compact(q, s->d_stage, elem, at, n, region, blocked_dst, s->stream);
        PRL_HIP_TRY(hipGetLastError());
        PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return PRL_OK;
}
// the blocked hands' value of the running average, per local column (see prl_solver::blk_avg)
static void blocked_avg_fill(const prl_solver* s, double* fill) {
    for (int j = 0; j < s->ncb; ++j) {
        const int A = board_col_actions(s, j);
        if (s->variant == PRL_CFR_PLUS) fill[j] = s->blk_avg[A < 4 ? A : 0];
        else fill[j] = s->board_avg_f64 ? 1.0 / (double)A : 0.0;  // prl_k_fhp_avg_from_sum on an all-zero sum, once the first sums exist
    }
}
// CFRPlus.py:65-87 for a hand whose regrets are 0 for ever (its strategy is the uniform float32 one): what the board pass would have stored
static void blocked_avg_step(prl_solver* s, int mode, double m_old, double m_new) {
    if (!mode) return;
    for (int A = 1; A < 4; ++A) {
        const float unif = (float)(1.0 / (double)A);
        double a = mode == 2 ? m_old * s->blk_avg[A] + m_new * (double)unif : (double)unif;
        if (s->avg_f32) a = (double)(float)a;
        s->blk_avg[A] = a;
    }
}
static void cfr_plus_weights(const prl_solver* s, int iter, int* mode, double* m_old, double* m_new) {
    *mode = 0; *m_old = 0.; *m_new = 0.;
    if (s->variant != PRL_CFR_PLUS) return;
    if (iter > s->delay) {  // CFRPlus.py:65-87: float64 weights from integer sums
        long long cw = 0;
        for (int k = s->delay + 1; k <= iter; ++k) cw += k;
        long long nw = iter - s->delay + 1;
        *m_old = (double)cw / (double)(cw + nw);
        *m_new = (double)nw / (double)(cw + nw);
        *mode = 2;
    } else if (iter == s->delay) *mode = 1;
}

int alloc_node_vectors(prl_solver* s, PrlDevState* st, bool with_br_idx) {
    const size_t nv = (size_t)s->T.n_nodes * 2 * s->T.R;
    TRY(dev_alloc(s, &st->reach, nv));
    TRY(dev_alloc(s, &st->ev, nv));
    TRY(dev_alloc(s, &st->ev_br, nv));
    TRY(dev_alloc(s, &st->expl, 2));
    st->br_idx = nullptr;
    if (with_br_idx) {
        TRY(dev_alloc(s, &st->br_idx, (size_t)s->T.n_nodes * s->T.R));
        PRL_HIP_TRY(hipMemsetAsync(st->br_idx, 0, (size_t)s->T.n_nodes * s->T.R * sizeof(int32_t), s->stream));  // terminal / chance rows stay 0
    }
    PRL_HIP_TRY(hipMemsetAsync(st->reach, 0, nv * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(st->ev, 0, nv * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(st->ev_br, 0, nv * sizeof(float), s->stream));
    return PRL_OK;
}

// generalised chance / equity constants, computed in float64 then rounded, as NumPy does
float chance_prob_f32(int n_children, int n_cards, int n_hole, int n_dealt) {
    double denom = (double)n_children * (double)prl_comb(n_cards - 2 * n_hole, n_dealt) / (double)prl_comb(n_cards, n_dealt);
    return (float)(1.0 / denom);
}
float eq_const_f32(int n_cards, int n_hole) {
    return (float)((double)prl_comb(n_cards, n_hole) / (double)prl_comb(n_cards - n_hole, n_hole));
}

int do_update_reach(prl_solver* s, const PrlDevState& st) {
    bool& root_set = &st == &s->S ? s->root_reach_set[0] : s->root_reach_set[1];  // (any other state: the scratch state of the average's evaluation)
    prl_launch_reach(s->T, st, s->ft.level_start.data(), s->stream, root_set && &st == &s->S);
    root_set = true;
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

// FUSED Vanilla / Linear: the board pass maintains avg_sum only; readers of the average call this first
static int ensure_board_avg(prl_solver* s) {
    if (!s->fused || !s->board_avg_stale) return PRL_OK;
    if (s->streets) {
        for (int lv = 0; lv < s->st.n_groups; ++lv) {
            PrlStParams q = s->sp;
            q.n_inst = s->st.group[lv].n_inst; q.col_base = s->st.group[lv].col_base; q.avg_sum = s->S.avg_sum; q.avg = s->d_avg;
            prl_launch_st_avg_from_sum(q, s->st.group[lv].spec, s->stream);
        }
        PRL_HIP_TRY(hipGetLastError());
        s->board_avg_stale = false;
        return PRL_OK;
    }
    PrlFhpParams p = s->fp;
    p.avg_sum = s->S.avg_sum + s->board_ofs;
    p.avg = s->d_avg + s->board_ofs;
    prl_launch_fhp_avg_from_sum(p, s->stream);
    PRL_HIP_TRY(hipGetLastError());
    s->board_avg_stale = false;
    return PRL_OK;
}

// STREETS (prl_st.h): the sweep over the streets that takes the place of the board pass: reach down street by street, the last
// street's pass, the other streets' passes bottom-up, the canonical sum over the first deal's outcomes into the trunk's chance leaves
int street_sweep(prl_solver* s, const PrlDevState& st, int mode, int src0, int src1, const double* strat_arr, const float* strat32) {
    PrlStParams p = s->sp;
    p.regret = strat32 ? const_cast<float*>(strat32) : s->d_regret;
    p.avg = s->avg_f32 ? nullptr : s->d_avg;  // (PRL_SOLVER_AVG_F32: the street columns' average lives in d_avg32; d_avg holds the trunk's float64 columns only)
    p.avg32 = s->d_avg32;
    p.avg_sum = s->S.avg_sum;
    p.strat_arr = strat_arr;
    p.variant = s->variant;
    p.iter = s->iter;
    p.avg_mode = s->fp.avg_mode; p.m_old = s->fp.m_old; p.m_new = s->fp.m_new;  // set by iteration_core for the update passes
    p.avgsum_mask = 0;
    if (&st == &s->S && strat_arr == nullptr) {  // pending Vanilla / Linear average updates ride on the reach walk of that seat
        const bool walks[2] = {prl_fhp_runs_seat(mode, 1), prl_fhp_runs_seat(mode, 0)};
        for (int q = 0; q < 2; ++q)
            if (walks[q] && s->avg_pending[q] >= 0) {
                p.avgsum_mask |= 1 << q;
                p.avgsum_iter[q] = s->avg_pending[q];
                s->avg_pending[q] = -1;
                s->board_avg_f64 = true;
                s->board_avg_stale = true;
            }
    }
    const int NG = s->st.n_groups, R = p.R, NL0 = s->n_trunk_leaves;
    const int width = prl_fhp_out_width(mode);
    auto level_params = [&](int g) {
        PrlStParams q = p;
        const PrlStLevelHost& H = s->st.group[g];
        q.n_inst = H.n_inst; q.col_base = H.col_base; q.inst = s->st_dev[g].inst;
        q.parent_reach = H.street == 0 ? st.reach : s->st_str[H.street - 1].leaf_reach;  // (node-major [n_nodes][2][R]: prl_vidx)
        q.leaf_reach = s->st_str[H.street].leaf_reach;
        q.child_val = s->st_str[H.street + 1].val;  // (the next street's rows: instances of any group and run-out chain roots)
        q.child_w = width;
        q.val = s->st_str[H.street].val;
#ifdef PRL_ST_TIMING
        if (H.last == (getenv("PRL_ST_TIMING_INNER") != nullptr)) q.timing = nullptr;  // instrumented builds clock the last street's pass, or the others'
#else
        q.timing = nullptr;
#endif
        return q;
    };
    // the groups [g0, g1) of one street, one launch after the other on the solver's stream (side by side on auxiliary streams was measured: the forks and
    // joins cost more than the small launches gain, 87 -> 82 M node-updates/s on DiscretizedNLHoldem 16 x 8 x 8, profiles/r93_group_streams_ab.txt)
    auto side_by_side = [&](int g0, int g1, auto&& launch) -> int {
        for (int g = g0; g < g1; ++g) {
            const int e = launch(g, s->stream);
            if (e) return e;
        }
        return PRL_OK;
    };
    auto street_end = [&](int g0) { int g1 = g0; while (g1 < NG && s->st.group[g1].street == s->st.group[g0].street) ++g1; return g1; };
    for (int g0 = 0; g0 < NG;) {  // reach down the streets
        const int g1 = street_end(g0);
        if (!s->st.group[g0].last) {
            const int e = side_by_side(g0, g1, [&](int g, hipStream_t on) { return prl_launch_st_down(s->st.group[g].spec, level_params(g), src0, src1, on); });
            if (e) { if (e != PRL_ERR_HIP) prl_set_error("street engine: unsupported strategy-source combination"); return e; }
        }
        g0 = g1;
    }
    if (s->n_chain) {  // the run-out chains below the all-in calls (prl_st.h): their values become rows of their streets before the passes read them
        PrlStChainIo io = {};
        for (int v = 0; v <= PRL_ST_MAX_LEVELS; ++v) {
            io.src[v] = v == 0 ? st.reach : s->st_str[v - 1].leaf_reach;
            io.val[v] = s->st_str[v].val;
        }
        if (s->chain_stream) {  // fork: the forest needs the leaf reach the DOWN launches left, nothing of the last street's passes
            PRL_HIP_TRY(hipEventRecord(s->chain_fork, s->stream));
            PRL_HIP_TRY(hipStreamWaitEvent(s->chain_stream, s->chain_fork, 0));
        }
        prl_launch_st_chain_eval(s->Tc, s->d_chain_terms, s->d_chain_bundles, s->n_chain_bundles, io, s->chain_dev, s->d_chain_ev, s->chain_level_start.data(), mode,
                                 s->chain_stream ? s->chain_stream : s->stream);
        if (s->chain_stream) PRL_HIP_TRY(hipEventRecord(s->chain_join, s->chain_stream));
    }
    bool chain_joined = !(s->n_chain && s->chain_stream);
    for (int g1 = NG; g1 > 0;) {  // the passes, street by street from the deepest one
        int g0 = g1 - 1;
        while (g0 > 0 && s->st.group[g0 - 1].street == s->st.group[g1 - 1].street) --g0;
        if (s->st.group[g0].last) {  // the last street's passes: the dominant kernels, one after the other on the solver's stream (timed there)
            for (int g = g1 - 1; g >= g0; --g) {
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                const bool timed = s->time_passes;
                if (timed) {
                    PRL_HIP_TRY(hipEventCreate(&ev0));
                    PRL_HIP_TRY(hipEventCreate(&ev1));
                    PRL_HIP_TRY(hipEventRecord(ev0, s->stream));
                }
                const int e = prl_launch_st_pass(s->st.group[g].spec, true, level_params(g), mode, src0, src1, s->stream);
                if (e) { prl_set_error("street engine: unsupported pass mode / strategy-source combination"); return e; }
                if (timed) {
                    PRL_HIP_TRY(hipEventRecord(ev1, s->stream));
                    s->pass_events.push_back(ev0);
                    s->pass_events.push_back(ev1);
                }
            }
        } else {
            if (!chain_joined) {  // join: a pass that is not on the last street sums rows of the forest's roots
                PRL_HIP_TRY(hipStreamWaitEvent(s->stream, s->chain_join, 0));
                chain_joined = true;
            }
            const int e = side_by_side(g0, g1, [&](int g, hipStream_t on) { return prl_launch_st_pass(s->st.group[g].spec, false, level_params(g), mode, src0, src1, on); });
            if (e) { if (e != PRL_ERR_HIP) prl_set_error("street engine: unsupported pass mode / strategy-source combination"); return e; }
        }
        g1 = g0;
    }
    if (!chain_joined) PRL_HIP_TRY(hipStreamWaitEvent(s->stream, s->chain_join, 0));
    // street-1 rows are outcome-major: [n_top][n_leaves][width][R] -> one canonical sum over the outcomes for all leaves at once
    const int W = NL0 * width * R, n_top = s->st.n_top;
    float* summed = s->d_row_sum;
    const float* rows = s->st_str[0].val;
    if (!s->exchange) prl_launch_fhp_chance_sum(rows, n_top, W, s->d_sum_scratch, summed, s->stream);
    else {
        const size_t per_rank = (size_t)s->n_units * W;
        prl_launch_fhp_chance_partial(rows, n_top, s->xlevel, W, s->d_sum_scratch, s->d_xlocal, s->stream);
        if (!s->exchange_async) PRL_HIP_TRY(hipStreamSynchronize(s->stream));
        ++s->n_exchanges;
        if (s->exchange(s->exchange_user, s->d_xlocal, s->d_xgather, (uint64_t)(per_rank * sizeof(float))) != 0) {
            prl_set_error("sharded solve: the exchange callback failed");
            return PRL_ERR_STATE;
        }
        prl_launch_fhp_chance_finish(s->d_xgather, s->n_units_all, s->xlevel, W, s->d_sum_scratch, summed, s->stream);
    }
    // the pieces go to their places in the trunk's chance leaves (as fused_board_pass does for its one chance node)
    const bool both = prl_fhp_runs_seat(mode, 0) && prl_fhp_runs_seat(mode, 1);
    const int seat = prl_fhp_runs_seat(mode, 0) ? 0 : 1;
    const bool with_br = prl_fhp_with_br(mode);
    PrlStScatter sc = {};
    sc.n_vec = width;
    auto add = [&](int src, int arr, int dst_seat) { sc.src_vec[sc.n_dst] = src; sc.dst_arr[sc.n_dst] = arr; sc.dst_seat[sc.n_dst] = dst_seat; ++sc.n_dst; };
    if (both) {
        add(0, 0, 0); add(1, 0, 1);
        add(with_br ? 2 : 0, 1, 0); add(with_br ? 3 : 1, 1, 1);
    } else if (mode == PRL_FHP_UPDATE1_EVAL1) {
        add(0, 0, 1); add(0, 1, 1);
        add(1, 2, 0); add(2, 2, 1);  // (seat 1 value under the new strategy, its best response): applied after the trunk update
    } else {
        add(0, 0, seat); add(with_br ? 1 : 0, 1, seat);
    }
    prl_launch_st_scatter_trunk(summed, s->d_trunk_leaves, NL0, R, sc, st.ev, st.ev_br, s->d_half, s->stream);
    if (!both && mode != PRL_FHP_UPDATE1_EVAL1 && seat == 0 && with_br && s->have_half && &st == &s->S)
        prl_launch_st_half_to_trunk(s->d_half, s->d_trunk_leaves, NL0, R, st.ev, st.ev_br, s->stream);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

// FUSED: board pass + canonical chance sum into the trunk's chance node (st = trunk state to read reach from / write to)
int fused_board_pass(prl_solver* s, const PrlDevState& st, int mode, int src0, int src1, const double* strat_arr, const float* strat32 = nullptr) {
    if (s->streets) return street_sweep(s, st, mode, src0, src1, strat_arr, strat32);
    PrlFhpParams p = s->fp;
    // the board regions of the column arrays (sorted storage, prl_fhp.h)
    // PRL_SRC_STRAT32: float32 strategy columns in the regret array's layout (read only)
    p.regret = (strat32 ? const_cast<float*>(strat32) : s->d_regret) + s->board_ofs;
    p.iter = s->iter;
    p.variant = s->variant;
    p.chance_reach = st.reach + prl_vidx(s->T, s->chance_trunk, 0);
    p.strat_arr = strat_arr ? strat_arr + s->board_ofs : nullptr;
    // pending Vanilla / Linear average updates ride on the phase-B walk of that seat (the training state only)
    p.avg_sum = s->S.avg_sum ? s->S.avg_sum + s->board_ofs : nullptr;
    p.avg = s->avg_f32 ? nullptr : s->d_avg + s->board_ofs;
    p.avg32 = s->d_avg32 ? s->d_avg32 + s->board_ofs : nullptr;
    p.avgsum_mask = 0;
    // level 0 of the canonical chance sum inside the pass (one row per 32-board block leaves the chip) whenever whole blocks are
    // what comes next: always without an exchange, and with one when the units exchanged are blocks or groups of blocks
    p.block_sum = (s->block_sum && (!s->exchange || s->xlevel >= 1)) ? 1 : 0;
    if (&st == &s->S && strat_arr == nullptr) {
        const bool walks[2] = {prl_fhp_runs_seat(mode, 1), prl_fhp_runs_seat(mode, 0)};  // seat q is the opponent of a batch of 1 - q
        for (int q = 0; q < 2; ++q)
            if (walks[q] && s->avg_pending[q] >= 0) {
                p.avgsum_mask |= 1 << q;
                p.avgsum_iter[q] = s->avg_pending[q];
                s->avg_pending[q] = -1;
                s->board_avg_f64 = true;
                s->board_avg_stale = true;
            }
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (s->time_passes) {
        PRL_HIP_TRY(hipEventCreate(&ev0));
        PRL_HIP_TRY(hipEventCreate(&ev1));
        PRL_HIP_TRY(hipEventRecord(ev0, s->stream));
    }
    int e = prl_launch_fhp_pass(p, mode, src0, src1, s->stream);
    if (e) { prl_set_error("fused engine: unsupported strategy-source combination"); return e; }
    if (s->time_passes) {
        PRL_HIP_TRY(hipEventRecord(ev1, s->stream));
        s->pass_events.push_back(ev0);
        s->pass_events.push_back(ev1);
    }
    // one canonical sum over the boards' rows (every vector of the row at once), then the pieces go to their places in the
    // trunk's chance node: value(s) -> ev, best response(s) -> ev_br (UPDATE1_EVAL1: kept in d_half until the trunk is updated)
    const bool both = prl_fhp_runs_seat(mode, 0) && prl_fhp_runs_seat(mode, 1);
    const int seat = prl_fhp_runs_seat(mode, 0) ? 0 : 1;
    const bool with_br = prl_fhp_with_br(mode);
    const int W = prl_fhp_out_width(mode) * p.R;
    float* summed = s->d_row_sum;
    const int n_blk = (p.n_boards + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
    if (!s->exchange) {
        if (p.block_sum) prl_launch_fhp_chance_finish(s->d_board_out, n_blk, 1, W, s->d_sum_scratch, summed, s->stream);
        else prl_launch_fhp_chance_sum(s->d_board_out, p.n_boards, W, s->d_sum_scratch, summed, s->stream);
    } else {
        // local units -> all-gather -> remaining levels over all units in global order (header: prl_solver_create_sharded)
        const size_t per_rank = (size_t)s->n_units * W;
        if (p.block_sum) prl_launch_fhp_chance_partial_from_blocks(s->d_board_out, n_blk, s->xlevel, W, s->d_xlocal, s->stream);
        else prl_launch_fhp_chance_partial(s->d_board_out, p.n_boards, s->xlevel, W, s->d_sum_scratch, s->d_xlocal, s->stream);
        if (!s->exchange_async) PRL_HIP_TRY(hipStreamSynchronize(s->stream));
        ++s->n_exchanges;
        if (s->exchange(s->exchange_user, s->d_xlocal, s->d_xgather, (uint64_t)(per_rank * sizeof(float))) != 0) {
            prl_set_error("sharded solve: the exchange callback failed");
            return PRL_ERR_STATE;
        }
        // rank-major blocks of whole units = global unit order already
        // (a shorter last shard: its units end before the zero padding of its block, which is therefore never read)
        prl_launch_fhp_chance_finish(s->d_xgather, s->n_units_all, s->xlevel, W, s->d_sum_scratch, summed, s->stream);
    }
    if (s->symmetrize) {  // suit isomorphism: the hand's value at the chance node = the mean over its suit orbit of the weighted sum (include/pokerrl_hip.h)
        prl_launch_fhp_symmetrize(summed, prl_fhp_out_width(mode), p.R, s->d_sym_class_of, s->d_sym_class_start, s->d_sym_class_hands, s->d_row_sym, s->stream);
        summed = s->d_row_sym;
    }
    const size_t vec = (size_t)p.R * sizeof(float);
    float* ch_ev = st.ev + prl_vidx(s->T, s->chance_trunk, 0);
    float* ch_br = st.ev_br + prl_vidx(s->T, s->chance_trunk, 0);
    auto put = [&](float* dst, int k, int n_vec) { return hipMemcpyAsync(dst, summed + (size_t)k * p.R, n_vec * vec, hipMemcpyDeviceToDevice, s->stream); };
    if (both) {
        PRL_HIP_TRY(put(ch_ev, 0, 2));
        PRL_HIP_TRY(put(ch_br, with_br ? 2 : 0, 2));
    } else if (mode == PRL_FHP_UPDATE1_EVAL1) {
        PRL_HIP_TRY(put(ch_ev + p.R, 0, 1));
        PRL_HIP_TRY(put(ch_br + p.R, 0, 1));
        PRL_HIP_TRY(put(s->d_half, 1, 2));  // (seat 1 value under the new strategy, its best response): applied after the trunk update
    } else {
        PRL_HIP_TRY(put(ch_ev + (size_t)seat * p.R, 0, 1));
        PRL_HIP_TRY(put(ch_br + (size_t)seat * p.R, with_br ? 1 : 0, 1));
        if (seat == 0 && with_br && s->have_half && &st == &s->S) {
            // seat 1's half of the same iterate (its value and best response under the current strategies, left by the pass
            // that updated it) completes the chance node: the trunk evaluation that follows yields both exploitabilities
            PRL_HIP_TRY(hipMemcpyAsync(ch_ev + p.R, s->d_half, vec, hipMemcpyDeviceToDevice, s->stream));
            PRL_HIP_TRY(hipMemcpyAsync(ch_br + p.R, s->d_half + p.R, vec, hipMemcpyDeviceToDevice, s->stream));
        }
    }
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int do_compute_ev(prl_solver* s, const PrlDevState& st, int fused_mode = PRL_FHP_EVAL) {
    // the history slot this evaluation writes (expl_to_history) is consumed HERE, on every exit path: a failed board pass / exchange must
    // not leave a pointer into a history block that ensure_hist may since have replaced
    float* const expl_dst = &st == &s->S ? s->expl_copy_dst : nullptr;
    if (&st == &s->S) s->expl_copy_dst = nullptr;
    if (s->fused) {
        const double* arr = nullptr;
        int s0 = s->src[0], s1 = s->src[1];
        const float* arr32 = nullptr;
        if (s->user_strategy_f64 == 1) {
            arr = s->d_user_strategy;
            s0 = s1 = PRL_SRC_ARR64;
        } else if (s->user_strategy_f64 == 0) {
            // exact best response of an explicit float32 strategy (LocalBRMaster.py:67-80): one evaluation pass that streams the
            // strategy like regrets (LDS prefetch), plays it as is -- no regret matching, no regret / average traffic
            arr32 = s->d_user_strategy32;
            s0 = s1 = PRL_SRC_STRAT32;
        }
        TRY(fused_board_pass(s, st, fused_mode, s0, s1, arr, arr32));
    }
    prl_launch_ev(s->T, st, s->ft.level_start.data(), s->d_term_nodes, s->n_term, s->stream, expl_dst);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int ensure_ev(prl_solver* s) {
    if (s->ev_valid) return PRL_OK;
    TRY(do_compute_ev(s, s->S));
    s->ev_valid = true;
    return PRL_OK;
}

int ensure_hist(prl_solver* s, int need) {
    if (need <= s->hist_cap) return PRL_OK;
    int cap = s->hist_cap ? s->hist_cap : 1024;
    while (cap < need) cap *= 2;
    float* q = nullptr;
    TRY(dev_alloc(s, &q, (size_t)cap * 2));
    if (s->d_expl_hist && s->hist_cap) PRL_HIP_TRY(hipMemcpyAsync(q, s->d_expl_hist, (size_t)s->hist_cap * 2 * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    s->d_expl_hist = q;  // the old block stays in `allocs` and is released with the solver
    s->hist_cap = cap;
    return PRL_OK;
}

// the evaluation that follows writes its exploitability into slot `iter` of the history itself (no separate copy)
int expl_to_history(prl_solver* s) {
    TRY(ensure_hist(s, s->iter + 1));
    s->expl_copy_dst = s->d_expl_hist + (size_t)s->iter * 2;
    return PRL_OK;
}

int record_expl(prl_solver* s) {
    TRY(ensure_hist(s, s->iter + 1));
    PRL_HIP_TRY(hipMemcpyAsync(s->d_expl_hist + (size_t)s->iter * 2, s->S.expl, 2 * sizeof(float), hipMemcpyDeviceToDevice, s->stream));
    return PRL_OK;
}

// trunk view of a Flop5Holdem-shaped tree: nodes dealt before the board, the chance node as a leaf
void make_trunk(const PrlFlatTree& t, int chance_node, int first_board_node, int board_nodes_total, PrlFlatTree* out, int* chance_trunk) {
    PrlFlatTree& u = *out;
    u = PrlFlatTree();
    u.rules = t.rules;
    u.game = t.game;
    std::vector<int> map(t.n_nodes, -1);
    for (int i = 0; i < t.n_nodes; ++i) {
        if (i >= first_board_node && i < first_board_node + board_nodes_total) continue;
        map[i] = u.n_nodes++;
        const bool is_ch = i == chance_node;
        u.kind.push_back(is_ch ? PRL_NODE_LEAF : t.kind[i]);
        u.actor.push_back(t.actor[i]);
        u.parent.push_back(t.parent[i] < 0 ? -1 : map[t.parent[i]]);
        u.child_idx.push_back(t.child_idx[i]);
        u.action.push_back(t.action[i]);
        u.acted_last.push_back(t.acted_last[i]);
        u.round.push_back(t.round[i]);
        u.board_id.push_back(-1);
        u.main_pot.push_back(t.main_pot[i]);
        u.depth.push_back(t.depth[i]);
        u.n_children.push_back(is_ch ? 0 : t.n_children[i]);
        u.first_col.push_back(t.first_col[i]);
        u.subtree_size.push_back(1);
    }
    *chance_trunk = map[chance_node];
    u.n_cols = 0;
    for (int i = 0; i < u.n_nodes; ++i)
        if (u.kind[i] == PRL_NODE_DECISION) u.n_cols += u.n_children[i];
    for (int c = 0; c < u.n_cols; ++c) { u.col_action.push_back(t.col_action[c]); u.col_node.push_back(map[t.col_node[c]]); }
    u.n_boards = 0;
    u.board_len = t.board_len;
    u.child_start.assign(u.n_nodes + 1, 0);
    for (int i = 0; i < u.n_nodes; ++i) u.child_start[i + 1] = u.child_start[i] + u.n_children[i];
    u.child_list.assign(u.child_start[u.n_nodes], -1);
    int max_depth = 0;
    for (int i = 1; i < u.n_nodes; ++i) {
        u.child_list[u.child_start[u.parent[i]] + u.child_idx[i]] = i;
        max_depth = max_depth > u.depth[i] ? max_depth : u.depth[i];
    }
    u.n_levels = max_depth + 1;
    u.level_start.assign(u.n_levels + 1, 0);
    for (int i = 0; i < u.n_nodes; ++i) u.level_start[u.depth[i] + 1]++;
    for (int d = 0; d < u.n_levels; ++d) u.level_start[d + 1] += u.level_start[d];
    u.level_nodes.assign(u.n_nodes, 0);
    std::vector<int32_t> fill(u.level_start.begin(), u.level_start.end() - 1);
    for (int i = 0; i < u.n_nodes; ++i) u.level_nodes[fill[u.depth[i]]++] = i;
}

// trunk view of a multi-street tree (STREETS engine): the nodes above every chance node, the chance nodes as leaves, action
// columns renumbered to the engine's internal order (trunk columns first)
void make_trunk_streets(const PrlFlatTree& t, const PrlStPlanHost& P, PrlFlatTree* out, std::vector<int32_t>* leaf_trunk_ids) {
    PrlFlatTree& u = *out;
    u = PrlFlatTree();
    u.rules = t.rules;
    u.game = t.game;
    std::vector<int> map(t.n_nodes, -1);
    for (int i = 0; i < t.n_nodes;) {
        map[i] = u.n_nodes++;
        const bool is_ch = t.kind[i] == PRL_NODE_CHANCE;
        u.kind.push_back(is_ch ? PRL_NODE_LEAF : t.kind[i]);
        u.actor.push_back(t.actor[i]);
        u.parent.push_back(t.parent[i] < 0 ? -1 : map[t.parent[i]]);
        u.child_idx.push_back(t.child_idx[i]);
        u.action.push_back(t.action[i]);
        u.acted_last.push_back(t.acted_last[i]);
        u.round.push_back(t.round[i]);
        u.board_id.push_back(-1);
        u.main_pot.push_back(t.main_pot[i]);
        u.depth.push_back(t.depth[i]);
        u.n_children.push_back(is_ch ? 0 : t.n_children[i]);
        u.first_col.push_back(P.trunk_col_of_node[i]);
        u.subtree_size.push_back(1);
        if (is_ch) { leaf_trunk_ids->push_back(map[i]); i += t.subtree_size[i]; }
        else ++i;
    }
    u.n_cols = P.n_trunk_cols;
    for (int c = 0; c < u.n_cols; ++c) { u.col_action.push_back(t.col_action[P.col_dfs[c]]); u.col_node.push_back(map[t.col_node[P.col_dfs[c]]]); }
    u.n_boards = 0;
    u.board_len = t.board_len;
    u.child_start.assign(u.n_nodes + 1, 0);
    for (int i = 0; i < u.n_nodes; ++i) u.child_start[i + 1] = u.child_start[i] + u.n_children[i];
    u.child_list.assign(u.child_start[u.n_nodes], -1);
    int max_depth = 0;
    for (int i = 1; i < u.n_nodes; ++i) {
        u.child_list[u.child_start[u.parent[i]] + u.child_idx[i]] = i;
        max_depth = max_depth > u.depth[i] ? max_depth : u.depth[i];
    }
    u.n_levels = max_depth + 1;
    u.level_start.assign(u.n_levels + 1, 0);
    for (int i = 0; i < u.n_nodes; ++i) u.level_start[u.depth[i] + 1]++;
    for (int d = 0; d < u.n_levels; ++d) u.level_start[d + 1] += u.level_start[d];
    u.level_nodes.assign(u.n_nodes, 0);
    std::vector<int32_t> fill(u.level_start.begin(), u.level_start.end() - 1);
    for (int i = 0; i < u.n_nodes; ++i) u.level_nodes[fill[u.depth[i]]++] = i;
}

}  // namespace

bool prl_fhp_shape_compiled(int shape_id);  // prl_fhp_kernels.hip

int prl_fhp_match_shape(const PrlFlatTree& t, int* chance_node, int* first_board_node, int* col_base, float* pots) {
    if (t.rules.n_hole_cards != 2 || t.rules.n_cards != 52 || t.board_len != 5) return -1;
    int ch = -1;
    for (int i = 0; i < t.n_nodes; ++i)
        if (t.kind[i] == PRL_NODE_CHANCE) {
            if (ch >= 0) return -1;  // exactly one chance node
            ch = i;
        }
    if (ch < 0 || t.n_children[ch] != t.n_boards) return -1;
    const int first = ch + 1;
    for (int i = 0; i < first; ++i)
        if (t.kind[i] == PRL_NODE_DECISION && t.first_col[i] >= t.first_col[first]) return -1;  // trunk columns precede
    for (int sid = 0; sid < PRL_FHP_N_SHAPES; ++sid) {
        if (!prl_fhp_shape_compiled(sid)) continue;
        const PrlFhpShapeDesc& d = prl_fhp_shape_desc(sid);
        const int N = d.n_nodes;
        if (first + (long long)t.n_boards * N != t.n_nodes) continue;  // the board subtrees are the tail of the DFS order
        bool ok = true;
        for (int b = 0; b < t.n_boards && ok; ++b) {
            const int base = first + b * N;
            if (t.board_id[base] != b) { ok = false; break; }
            if (b > 0 && b < t.n_boards - 1) continue;  // subtrees are replicas by construction; check the first and the last
            for (int n = 0; n < N && ok; ++n) {
                const int g = base + n;
                if (t.kind[g] != d.kind[n] || t.n_children[g] != d.nch[n]) ok = false;
                else if (t.kind[g] == PRL_NODE_DECISION && t.actor[g] != d.actor[n]) ok = false;
                else if (n > 0 && t.parent[g] != base + d.parent[n]) ok = false;
                else if (t.kind[g] == PRL_NODE_TERM_FOLD && t.acted_last[g] != d.folder[n]) ok = false;
                else if (t.kind[g] == PRL_NODE_DECISION && t.first_col[g] != t.first_col[base] + d.col0[n]) ok = false;
            }
        }
        if (!ok) continue;
        *chance_node = ch;
        *first_board_node = first;
        *col_base = t.first_col[first];
        for (int n = 0; n < N; ++n) pots[n] = (float)t.main_pot[first + n];
        return sid;
    }
    return -1;
}

extern "C" {

static int32_t solver_create_impl(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t world, int32_t rank,
                                  prl_exchange_fn exchange, void* exchange_user, prl_solver_t** out, int64_t shard_boards = 0, int64_t total_boards = 0,
                                  const void* rccl_uid = nullptr, int32_t flags = 0, const int32_t* board_mult = nullptr, int32_t symmetrize = 0) {
    if (!tree || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (variant < 0 || variant > 2 || delay < 0 || engine < 0 || engine > 2) { prl_set_error("bad variant / delay / engine"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device: the solver has no CPU fallback"); return PRL_ERR_NO_DEVICE; }
    const PrlFlatTree& full = *prl_tree_flat(tree);
    if (full.is_partial) { prl_set_error("partial tree (stop_at_street): structure only, there is nothing to solve below the cut"); return PRL_ERR_UNSUPPORTED; }
    const PrlRules& r = full.rules;
    if (r.n_hole_cards == 1 && (r.range_size > 128 || full.board_len != 1)) { prl_set_error("1-card games: R <= 128, 1 board card"); return PRL_ERR_UNSUPPORTED; }
    if (r.n_hole_cards == 2 && (r.n_cards != 52 || r.n_suits != 4 || full.board_len != 5 || r.rank_rule != 2)) {
        prl_set_error("2-card games: 52-card deck with 5-card run-outs (Flop5Holdem; hold'em games dealing 3 + 1 + 1) only");
        return PRL_ERR_UNSUPPORTED;
    }
    for (int i = 0; i < full.n_nodes; ++i)
        if (full.kind[i] == PRL_NODE_DECISION && full.n_children[i] > 96) { prl_set_error("more than 96 actions at a node"); return PRL_ERR_UNSUPPORTED; }
    int ch_node = -1, first_board = -1, col_base = -1;
    float pots[PRL_FHP_MAX_NODES];
    const int shape_id = prl_fhp_match_shape(full, &ch_node, &first_board, &col_base, pots);
    const bool shape_ok = shape_id >= 0;
    bool fused = false, streets = false;
    // trees that deal on several streets (or whose one deal hangs below several chance nodes): the per-street fused engine (prl_st.h).
    // Its street-1 chance weight counts the GLOBAL number of first-deal outcomes in a sharded solve, known only below: the plan is
    // built here to see whether the engine applies and once more with that count.
    PrlStPlanHost st_plan;
    std::string st_why;
    const bool st_ok = !shape_ok && engine != PRL_ENGINE_LEVELS && r.n_hole_cards == 2 && prl_st_build(full, 0, &st_plan, &st_why) == PRL_OK;
    if (engine == PRL_ENGINE_FUSED) {
        if (!shape_ok && !st_ok) {
            prl_set_error("fused engine: neither a single-deal tree whose board subtree is a registered shape (prl_fhp.h) nor a multi-street tree of registered street shapes (prl_st.h)" +
                          (st_why.empty() ? std::string() : ": " + st_why));
            return PRL_ERR_UNSUPPORTED;
        }
        fused = true;
    } else if (engine == PRL_ENGINE_AUTO) fused = shape_ok || st_ok;
    streets = fused && !shape_ok;
    long long mult_sum = 0;  // weighted boards: the chance probability counts the boards the listed ones stand for
    if (board_mult) {
        if (!fused || streets || exchange) { prl_set_error("weighted boards: single-deal fused engine, one GPU"); return PRL_ERR_UNSUPPORTED; }
        for (int i = 0; i < full.n_boards; ++i) {
            if (board_mult[i] < 1) { prl_set_error("weighted boards: every multiplicity must be >= 1"); return PRL_ERR_ARG; }
            mult_sum += board_mult[i];
        }
        if (mult_sum > 0x7fffffffll) { prl_set_error("weighted boards: too many boards in all"); return PRL_ERR_ARG; }
        if (symmetrize) {
            // the orbit mean at the chance node equals the whole game's sum only for suit-class representatives with their orbit sizes -- any other
            // weighting (importance-sampled boards, ...) is `symmetrize` 0. Checked here: every listed board is the canonical member of its class
            // (prl_policy.h: prl_suit_canon; cards ascending), its multiplicity is the size of its orbit, and -- unless symmetrize is
            // PRL_SYMMETRIZE_SUBSET -- the list covers the game: the multiplicities add up to C(n_cards, k).
            if (symmetrize != 1 && symmetrize != PRL_SYMMETRIZE_SUBSET) { prl_set_error("symmetrize: 0, 1 (all suit classes) or 2 (a subset of them)"); return PRL_ERR_ARG; }
            const int n_perm = prl_n_suit_perms(r.n_suits);
            for (int i = 0; i < full.n_boards; ++i) {
                const int8_t* b = full.boards.data() + (size_t)i * full.board_len;
                int8_t cb[5];
                prl_suit_canon(b, full.board_len, r.n_suits, cb);
                int stab = 0;  // relabellings that leave the (canonical) board in place
                for (int k = 0; k < n_perm; ++k) {
                    int p[4] = {0, 1, 2, 3};
                    prl_suit_perm(k, r.n_suits, p);
                    unsigned long long m0 = 0ull, m1 = 0ull;
                    for (int c = 0; c < full.board_len; ++c) { m0 |= 1ull << b[c]; m1 |= 1ull << ((b[c] / r.n_suits) * r.n_suits + p[b[c] % r.n_suits]); }
                    stab += m0 == m1;
                }
                bool same = true;
                for (int c = 0; c < full.board_len; ++c) same = same && cb[c] == b[c];
                if (!same || board_mult[i] != n_perm / stab) {
                    prl_set_error("symmetrize: board " + std::to_string(i) + (same ? " carries a multiplicity that is not the size of its suit orbit" :
                                  " is not the representative of its suit class (the lexicographically smallest relabelling, cards ascending)") +
                                  " -- weighted boards that are not suit classes go with symmetrize = 0");
                    return PRL_ERR_ARG;
                }
            }
            if (symmetrize == 1 && mult_sum != prl_comb(r.n_cards, full.board_len)) {
                prl_set_error("symmetrize: the listed classes do not cover the game (their multiplicities add up to " + std::to_string(mult_sum) + " of " +
                              std::to_string(prl_comb(r.n_cards, full.board_len)) + " boards); a deliberate subset is symmetrize = 2");
                return PRL_ERR_ARG;
            }
        }
    } else if (symmetrize) { prl_set_error("symmetrize needs board multiplicities"); return PRL_ERR_ARG; }
    if ((flags & PRL_SOLVER_AVG_F32) && (!fused || variant != PRL_CFR_PLUS)) {
        prl_set_error("PRL_SOLVER_AVG_F32: the fused engines (single-deal board pass, per-street passes) with CFR+ only (the variant whose running average the passes blend)");
        return PRL_ERR_UNSUPPORTED;
    }
    if (exchange && !fused) { prl_set_error("sharded solve: FUSED engine only (board / street subtrees of registered shapes, prl_fhp.h, prl_st.h)"); return PRL_ERR_UNSUPPORTED; }
    // the unit a sharded solve splits: the outcomes of the first deal (boards of a single-deal tree, flops of a multi-street one)
    const int n_top = streets ? st_plan.n_top : full.n_boards;
    // shard geometry: equal shards unless told otherwise; ragged = every rank but the last holds shard_boards (whole canonical
    // units of the level exchanged), the last one the rest of total_boards
    if (shard_boards <= 0) { shard_boards = n_top; total_boards = (int64_t)n_top * world; }
    {
        const int64_t mine = rank == world - 1 ? total_boards - (int64_t)(world - 1) * shard_boards : shard_boards;
        if (mine <= 0 || mine > shard_boards || mine != n_top || total_boards > 0x7fffffffll) {
            prl_set_error("sharded solve: this rank's tree does not hold its share of the board list (ranks before the last: shard_boards; the last: the rest)");
            return PRL_ERR_ARG;
        }
    }

    prl_solver* s = new prl_solver();
    s->world = world;
    s->rank = rank;
    {   // FNV-1a over everything a checkpoint of this solver is tied to (prl_solver_load_state)
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(full.boards.data(), full.boards.size());
        // the meaningful fields only: array tails past n_rounds / n_bet_sizes are whatever the caller's memory held (both structs are all
        // 4- / 8-byte fields without padding); zero-initialised descriptions hash as before
        PrlGame g = full.game;
        PrlRules r = full.rules;
        for (int i = g.n_rounds < 0 ? 0 : g.n_rounds; i < 4; ++i) g.max_raises[i] = 0;
        for (int i = g.n_bet_sizes < 0 ? 0 : g.n_bet_sizes; i < PRL_MAX_BET_SIZES; ++i) g.bet_fracs[i] = 0.;
        for (int i = r.n_rounds < 0 ? 0 : r.n_rounds; i < 4; ++i) r.board_cards_in_round[i] = 0;
        static_assert(sizeof(PrlGame) == 18 * 4 + 8 * PRL_MAX_BET_SIZES && sizeof(PrlRules) == 13 * 4, "no padding in the hashed descriptions");
        mix(&g, sizeof(g));
        mix(&r, sizeof(r));
        const int32_t wr[2] = {world, rank};
        mix(wr, sizeof(wr));
        if (board_mult) {  // weighted boards: the multiplicities and the symmetrisation belong to the problem
            mix(board_mult, sizeof(int32_t) * (size_t)full.n_boards);
            mix(&symmetrize, sizeof(symmetrize));
        }
        s->fingerprint = h;
    }
    s->exchange = exchange;
    s->exchange_user = exchange_user;
    s->variant = variant;
    s->delay = delay;
    s->fused = fused;
    s->streets = streets;
    if (streets) {
        if (exchange && prl_st_build(full, total_boards, &st_plan, &st_why) != PRL_OK) { prl_set_error("street engine: " + st_why); delete s; return PRL_ERR_UNSUPPORTED; }
        s->st = st_plan;
        s->col_dfs = st_plan.col_dfs;
        bool identity = true;
        for (size_t c = 0; c < s->col_dfs.size() && identity; ++c) identity = s->col_dfs[c] == (int32_t)c;
        if (identity) s->col_dfs.clear();
    }
    s->sorted = fused && !streets;
    s->small_tree = !fused && r.n_hole_cards == 1 && (long long)full.n_nodes * r.range_size <= 32768 && !getenv("PRL_NO_SMALL_TREE");
    s->full_nodes = full.n_nodes;
    s->full_cols = full.n_cols;
    s->R = r.range_size;
    std::vector<int32_t> st_leaf_ids;
    if (streets) make_trunk_streets(full, s->st, &s->ft, &st_leaf_ids);
    else if (fused) make_trunk(full, ch_node, first_board, full.n_boards * prl_fhp_shape_desc(shape_id).n_nodes, &s->ft, &s->chance_trunk);
    else s->ft = full;
    std::vector<int32_t> chain_full_node;  // forest node -> node of the full tree (run-out chains of the per-street engine, prl_st.h)
    std::vector<int32_t> chain_root_id;    // per chain root (s->st.chain order, sorted by street below): its forest node
    std::vector<float> chain_w;            // forest node -> the weight of each of its outcomes (chance nodes)
    if (streets && !s->st.chain.empty()) {
        std::stable_sort(s->st.chain.begin(), s->st.chain.end(), [](const PrlStChainKid& a, const PrlStChainKid& b) { return a.street < b.street; });
        PrlFlatTree& u = s->chain_ft;
        u = PrlFlatTree();
        u.rules = full.rules; u.game = full.game; u.board_len = full.board_len; u.n_boards = full.n_boards;
        std::vector<int> map(full.n_nodes, -1);
        for (const PrlStChainKid& ck : s->st.chain) {
            chain_root_id.push_back(u.n_nodes);
            for (int i = ck.node; i < ck.node + full.subtree_size[ck.node]; ++i) {
                map[i] = u.n_nodes++;
                chain_full_node.push_back(i);
                u.kind.push_back(full.kind[i]); u.actor.push_back(-1);
                u.parent.push_back(i == ck.node ? -1 : map[full.parent[i]]);
                u.child_idx.push_back(full.child_idx[i]); u.action.push_back(full.action[i]); u.acted_last.push_back(full.acted_last[i]);
                u.round.push_back(full.round[i]); u.board_id.push_back(full.board_id[i]); u.main_pot.push_back(full.main_pot[i]);
                u.depth.push_back(full.depth[i] - full.depth[ck.node]);
                u.n_children.push_back(full.n_children[i]); u.first_col.push_back(-1); u.subtree_size.push_back(full.subtree_size[i]);
            }
        }
        u.child_start.assign(u.n_nodes + 1, 0);
        for (int i = 0; i < u.n_nodes; ++i) u.child_start[i + 1] = u.child_start[i] + u.n_children[i];
        u.child_list.assign(u.child_start[u.n_nodes] > 0 ? u.child_start[u.n_nodes] : 1, -1);
        int max_depth = 0;
        for (int i = 0; i < u.n_nodes; ++i) {
            if (u.parent[i] >= 0) u.child_list[u.child_start[u.parent[i]] + u.child_idx[i]] = i;
            max_depth = max_depth > u.depth[i] ? max_depth : u.depth[i];
        }
        u.n_levels = max_depth + 1;
        u.level_start.assign(u.n_levels + 1, 0);
        for (int i = 0; i < u.n_nodes; ++i) u.level_start[u.depth[i] + 1]++;
        for (int d = 0; d < u.n_levels; ++d) u.level_start[d + 1] += u.level_start[d];
        u.level_nodes.assign(u.n_nodes, 0);
        std::vector<int32_t> fill(u.level_start.begin(), u.level_start.end() - 1);
        for (int i = 0; i < u.n_nodes; ++i) u.level_nodes[fill[u.depth[i]]++] = i;
        s->n_chain = (int)s->st.chain.size();
    }
    const PrlFlatTree& ft = s->ft;
#define FAIL_IF(x) do { int e_ = (x); if (e_) { prl_solver_destroy(s); return e_; } } while (0)
    if (hipStreamCreate(&s->stream) != hipSuccess) { prl_set_error("hipStreamCreate failed"); delete s; return PRL_ERR_HIP; }
#if !defined(PRL_EMU)
    if (rccl_uid) {  // the library's own exchange: this rank joins the communicator of the solve (collective: every rank is here)
        ncclUniqueId id;
        memcpy(&id, rccl_uid, sizeof(id));
        ncclComm_t comm = nullptr;
        const ncclResult_t r = prl_rccl_api().CommInitRank(&comm, world, id, rank);
        if (r != ncclSuccess) {
            prl_set_error(std::string("ncclCommInitRank: ") + prl_rccl_api().GetErrorString(r) + " (RCCL bound from " + prl_rccl_api().path + ")");
            (void)hipStreamDestroy(s->stream);
            delete s;
            return PRL_ERR_HIP;
        }
        s->rccl_comm = comm;
        s->exchange = prl_rccl_exchange;
        s->exchange_user = s;
        s->exchange_async = true;
    }
#endif
    PrlDevTree& T = s->T;
    T.n_nodes = ft.n_nodes; T.n_cols = ft.n_cols; T.R = r.range_size; T.n_hole = r.n_hole_cards; T.n_cards = r.n_cards;
    T.n_suits = r.n_suits; T.rank_rule = r.rank_rule; T.n_boards = ft.n_boards; T.board_len = ft.board_len; T.n_levels = ft.n_levels;
    FAIL_IF(dev_upload(s, &T.kind, ft.kind));
    FAIL_IF(dev_upload(s, &T.actor, ft.actor));
    FAIL_IF(dev_upload(s, &T.parent, ft.parent));
    FAIL_IF(dev_upload(s, &T.child_idx, ft.child_idx));
    FAIL_IF(dev_upload(s, &T.acted_last, ft.acted_last));
    FAIL_IF(dev_upload(s, &T.board_id, ft.board_id));
    FAIL_IF(dev_upload(s, &T.main_pot, ft.main_pot));
    FAIL_IF(dev_upload(s, &T.n_children, ft.n_children));
    FAIL_IF(dev_upload(s, &T.first_col, ft.first_col));
    FAIL_IF(dev_upload(s, &T.child_start, ft.child_start));
    FAIL_IF(dev_upload(s, &T.child_list, ft.child_list));
    FAIL_IF(dev_upload(s, &T.level_nodes, ft.level_nodes));
    FAIL_IF(dev_upload(s, &T.boards, full.boards));
    std::vector<int16_t> hole((size_t)T.R * 2);
    std::vector<uint16_t> hole_packed((size_t)T.R);
    for (int h = 0; h < T.R; ++h) {
        int c1, c2;
        prl_hand_cards(r, h, &c1, &c2);
        hole[2 * h] = (int16_t)c1;
        hole[2 * h + 1] = (int16_t)c2;
        hole_packed[h] = (uint16_t)((c1 & 0xFF) | ((c2 & 0xFF) << 8));
    }
    FAIL_IF(dev_upload(s, &T.hole, hole));
    int n_chance_children = full.n_boards;
    for (int i = 0; i < full.n_nodes; ++i)
        if (full.kind[i] == PRL_NODE_CHANCE) { n_chance_children = full.n_children[i]; break; }
    {   // per chance node: the weight of each of its outcomes for a hand the outcome does not block. Generalised StrategyFiller.py:166
        // (SURVEY Appendix C): 1 / (n_children * C(N' - 2H, k) / C(N', k)), N' = cards not on the board yet, k = cards dealt here
        auto dealt = [&](int row) { int n = 0; if (row >= 0) for (int c = 0; c < full.board_len; ++c) n += full.boards[(size_t)row * full.board_len + c] >= 0; return n; };
        std::vector<float> w(ft.n_nodes, 0.f);
        bool first = true;
        for (int i = 0; i < full.n_nodes; ++i) {
            if (full.kind[i] != PRL_NODE_CHANCE) continue;
            const int before = dealt(full.board_id[i]);
            const int child = full.child_list[full.child_start[i]];
            const int k = dealt(full.board_id[child]) - before;
            // sharded (one chance node, fused engine): the GLOBAL number of boards
            const float p = chance_prob_f32(board_mult ? (int)mult_sum : exchange ? (int)total_boards : full.n_children[i], r.n_cards - before, r.n_hole_cards, k);
            if (first) { T.chance_prob = p; first = false; }
            if (!fused && i < ft.n_nodes) w[i] = p;
        }
        if (first) T.chance_prob = chance_prob_f32(board_mult ? (int)mult_sum : exchange ? (int)total_boards : n_chance_children, r.n_cards, r.n_hole_cards, full.board_len);
        FAIL_IF(dev_upload(s, &T.chance_w, w));
        if (s->n_chain) {  // the run-out chains' own chance nodes (the forest of the per-street engine)
            std::vector<float> wc(s->chain_ft.n_nodes, 0.f);
            for (int i = 0; i < s->chain_ft.n_nodes; ++i) {
                const int f = chain_full_node[i];
                if (full.kind[f] != PRL_NODE_CHANCE) continue;
                const int before = dealt(full.board_id[f]);
                const int k = dealt(full.board_id[full.child_list[full.child_start[f]]]) - before;
                wc[i] = chance_prob_f32(full.n_children[f], r.n_cards - before, r.n_hole_cards, k);
            }
            FAIL_IF(dev_upload(s, &s->Tc.chance_w, wc));
            chain_w = wc;
        }
    }
    T.eq_const = eq_const_f32(r.n_cards, r.n_hole_cards);

    std::vector<int32_t> term, np[2];
    for (int i = 0; i < ft.n_nodes; ++i) {
        if (ft.kind[i] == PRL_NODE_TERM_FOLD || ft.kind[i] == PRL_NODE_TERM_SHOWDOWN) term.push_back(i);
        if (ft.kind[i] == PRL_NODE_DECISION) np[ft.actor[i]].push_back(i);
    }
    s->n_term = (int)term.size();
    FAIL_IF(dev_upload(s, (const int32_t**)&s->d_term_nodes, term));
    for (int p = 0; p < 2; ++p) {
        s->n_nodes_p[p] = (int)np[p].size();
        FAIL_IF(dev_upload(s, (const int32_t**)&s->d_nodes_p[p], np[p]));
    }
    FAIL_IF(dev_upload(s, (const int32_t**)&s->d_col_node, ft.col_node));

    if (T.n_hole == 2) {  // showdown plans of every board + the "no board" plan, built on the device
        const int n_plans = full.n_boards + 1;
        T.plan_stride = T.R;
        T.cl_stride = T.n_cards * (T.n_cards - 1);
        int16_t *sh, *pos, *gs, *ge, *cl, *hgs, *hge, *pp = nullptr;
        uint32_t* clx = nullptr;
        int32_t *nl, *nd;
        FAIL_IF(dev_alloc(s, &sh, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &pos, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &gs, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &ge, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &cl, (size_t)n_plans * T.cl_stride));
        FAIL_IF(dev_alloc(s, &nl, (size_t)n_plans));
        FAIL_IF(dev_alloc(s, &nd, (size_t)n_plans));
        FAIL_IF(dev_alloc(s, &hgs, (size_t)n_plans * T.plan_stride));
        FAIL_IF(dev_alloc(s, &hge, (size_t)n_plans * T.plan_stride));
        if (fused) FAIL_IF(dev_alloc(s, &clx, (size_t)n_plans * PRL_CLX_WORDS));
        if (s->sorted) FAIL_IF(dev_alloc(s, &pp, (size_t)n_plans * PRL_PP_STRIDE));
        uint8_t* klh = nullptr;  // the LEVELS engine's showdown terminals (a fused solver's trunk has none)
        if (!fused || s->n_chain) FAIL_IF(dev_alloc(s, &klh, (size_t)n_plans * T.R * 4));  // (the run-out chains of the per-street engine end in showdown terminals too)
        PrlDevTree Tb = T;
        Tb.n_boards = full.n_boards;
        prl_launch_plan_build(Tb, n_plans, sh, pos, gs, ge, cl, nl, hgs, hge, clx, nd, klh, pp, s->stream);
        // the LEVELS kernels address plan `board_id`, or plan index T.n_boards for "no board"; in the FUSED engine the
        // trunk tree has n_boards == 0, so its plan pointers are based at the last (no-board) plan
        const size_t off = fused ? (size_t)full.n_boards : 0;
        T.plan_sh = sh + off * T.plan_stride; T.plan_pos = pos + off * T.plan_stride; T.plan_gs = gs + off * T.plan_stride;
        T.plan_ge = ge + off * T.plan_stride; T.plan_cl = cl + off * T.cl_stride; T.plan_nlive = nl + off; T.plan_ndealt = nd + off;
        T.plan_hgs = hgs + off * T.plan_stride; T.plan_hge = hge + off * T.plan_stride; T.plan_clx = clx ? clx + off * PRL_CLX_WORDS : nullptr;
        T.plan_klh = klh;
        if (s->n_chain) {  // the forest addresses the plans by board row, like a LEVELS tree
            const float* wc = s->Tc.chance_w;
            s->Tc = T;
            s->Tc.chance_w = wc;
            s->Tc.n_boards = full.n_boards;
            s->Tc.plan_sh = sh; s->Tc.plan_pos = pos; s->Tc.plan_gs = gs; s->Tc.plan_ge = ge; s->Tc.plan_cl = cl; s->Tc.plan_nlive = nl; s->Tc.plan_ndealt = nd;
            s->Tc.plan_hgs = hgs; s->Tc.plan_hge = hge; s->Tc.plan_clx = clx; s->Tc.plan_klh = klh;
        }
        if (streets) {
            PrlStParams& sp = s->sp;
            sp.R = T.R; sp.eq_const = T.eq_const; sp.n_cards = T.n_cards;
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            sp.max_grid = cus > 0 ? cus : 256;
            const char* g = getenv("PRL_FHP_GRID");
            if (g && atoi(g) > 0) sp.max_grid = atoi(g);
            sp.plan_stride = T.plan_stride; sp.cl_stride = T.cl_stride;
            sp.plan_pos = pos; sp.plan_hgs = hgs; sp.plan_hge = hge; sp.plan_cl = cl; sp.plan_clx = clx; sp.plan_nlive = nl; sp.plan_ndealt = nd;
            FAIL_IF(dev_upload(s, &sp.hole_packed, hole_packed));
        } else if (fused) {
            PrlFhpParams& fp = s->fp;
            fp.n_boards = full.n_boards; fp.R = T.R; fp.np = PRL_FHP_NP;
            if (T.R != PRL_PP_R) { prl_set_error("fused engine: 1326-hand ranges only"); prl_solver_destroy(s); return PRL_ERR_UNSUPPORTED; }
            {   // persistent workgroups, one per CU (LDS-bound occupancy): each walks its boards with the next one prefetching
                int dev = 0, cus = 256;
                if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                fp.max_grid = cus > 0 ? cus : 256;
            }
            {
                const char* g = getenv("PRL_FHP_GRID");
                if (g && atoi(g) > 0) fp.max_grid = atoi(g);
            }
            {   // level 0 of the chance sum inside the pass rounds a workgroup's board range up to whole 32-board blocks: with few
                // boards per workgroup that idles CUs (4096 boards on 256 CUs: 16 per workgroup -> 32 -> half the chip), so the
                // pass keeps per-board rows unless the rounding costs < ~6 %. PRL_FHP_BLOCK_SUM=0/1 forces either path (tests).
                const int g = full.n_boards < fp.max_grid ? full.n_boards : fp.max_grid;
                const int raw = (full.n_boards + g - 1) / g;
                const int rounded = (raw + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK * PRL_CHANCE_BLOCK;
                s->block_sum = raw >= PRL_CHANCE_BLOCK && (long long)rounded * 16 <= (long long)raw * 17;
                const char* f = getenv("PRL_FHP_BLOCK_SUM");
                if (f) s->block_sum = atoi(f) != 0;
                if (getenv("PRL_FHP_NO_BLOCK_SUM")) s->block_sum = false;
            }
            fp.no_steady = getenv("PRL_FHP_NO_STEADY") ? 1 : 0;  // tests: the generic pass in the steady state too
            fp.chance_prob = T.chance_prob; fp.eq_const = T.eq_const;
            {   // the boards' chance weights as an array (the pass reads a board's weight a board ahead, unconditionally)
                std::vector<float> bw((size_t)full.n_boards);
                for (int i = 0; i < full.n_boards; ++i) bw[(size_t)i] = board_mult ? T.chance_prob * (float)board_mult[i] : T.chance_prob;  // one float32 rounding, as the oracle's
                FAIL_IF(dev_upload(s, &s->d_board_w, bw));
                fp.board_w = s->d_board_w;
            }
            if (board_mult) {
                if (symmetrize) {
                    // classes of hands under the 24 suit permutations: (low rank, high rank, suited) -- 13 pairs x 6, 78 suited x 4, 78 offsuit x 12
                    const int R = r.range_size, NS = r.n_suits;
                    std::vector<int32_t> key((size_t)R), class_of((size_t)R), start, hands;
                    std::vector<int> order;
                    for (int h = 0; h < R; ++h) {
                        int c1, c2;
                        prl_hole_cards_2(h, r.n_cards, &c1, &c2);
                        const int r1 = prl_card_rank(c1, NS), r2 = prl_card_rank(c2, NS);
                        key[(size_t)h] = ((r1 < r2 ? r1 : r2) * r.n_ranks + (r1 < r2 ? r2 : r1)) * 2 + (prl_card_suit(c1, NS) == prl_card_suit(c2, NS) ? 1 : 0);
                    }
                    std::vector<int32_t> first_of((size_t)r.n_ranks * r.n_ranks * 2, -1);
                    int n_classes = 0;
                    for (int h = 0; h < R; ++h) {  // classes numbered by their first hand
                        if (first_of[(size_t)key[(size_t)h]] < 0) first_of[(size_t)key[(size_t)h]] = n_classes++;
                        class_of[(size_t)h] = first_of[(size_t)key[(size_t)h]];
                    }
                    start.assign((size_t)n_classes + 1, 0);
                    for (int h = 0; h < R; ++h) ++start[(size_t)class_of[(size_t)h] + 1];
                    for (int k = 0; k < n_classes; ++k) start[(size_t)k + 1] += start[(size_t)k];
                    hands.resize((size_t)R);
                    std::vector<int32_t> fill(start.begin(), start.end() - 1);
                    for (int h = 0; h < R; ++h) hands[(size_t)fill[(size_t)class_of[(size_t)h]]++] = h;  // ascending hand index inside a class
                    FAIL_IF(dev_upload(s, &s->d_sym_class_of, class_of));
                    FAIL_IF(dev_upload(s, &s->d_sym_class_start, start));
                    FAIL_IF(dev_upload(s, &s->d_sym_class_hands, hands));
                    s->symmetrize = true;
                }
            }
            const PrlFhpShapeDesc& sd = prl_fhp_shape_desc(shape_id);
            fp.shape = shape_id; fp.n_cols_board = sd.n_cols; fp.n_dec = sd.n_dec;
            for (int j = 0; j < sd.n_dec; ++j) { fp.dec_nch[j] = sd.dec_nch[j]; fp.dec_col0[j] = sd.dec_col0[j]; }
            for (int n = 0; n < sd.n_nodes; ++n) fp.pot[n] = pots[n];
            fp.plan_clx = clx; fp.plan_pp = pp;
            s->col_base = col_base; s->ncb = sd.n_cols;
            s->board_ofs = ((size_t)T.n_cols * T.R + 3) & ~(size_t)3;  // 16-byte aligned board region (float32 and float64 arrays alike)
            s->col_elems = s->board_ofs + (size_t)full.n_boards * sd.n_cols * PRL_FHP_NP;
            if (col_base != T.n_cols) { prl_set_error("fused engine: the board columns must follow the trunk's"); prl_solver_destroy(s); return PRL_ERR_UNSUPPORTED; }
        }
    }

    const size_t nc_full = col_array_elems(s);
    FAIL_IF(dev_alloc(s, &s->d_regret, nc_full));
    s->avg_f32 = (flags & PRL_SOLVER_AVG_F32) != 0;
    if (s->avg_f32) {  // float64 for the trunk's few columns (the LEVELS kernels keep the reference's dtype), float32 for the boards'
        FAIL_IF(dev_alloc(s, &s->d_avg, (size_t)T.n_cols * T.R));
        FAIL_IF(dev_alloc(s, &s->d_avg32, nc_full));
    } else FAIL_IF(dev_alloc(s, &s->d_avg, nc_full));
    s->S.regret = s->d_regret;
    s->S.avg = s->d_avg;
    FAIL_IF(dev_alloc(s, &s->S.strategy, (size_t)T.n_cols * T.R));  // FUSED: trunk columns only
    FAIL_IF(dev_alloc(s, &s->S.strat_f64, (size_t)T.n_nodes));
    FAIL_IF(dev_alloc(s, &s->S.avg_f64, (size_t)T.n_nodes));
    if (variant != PRL_CFR_PLUS) FAIL_IF(dev_alloc(s, &s->S.avg_sum, nc_full));
    FAIL_IF(alloc_node_vectors(s, &s->S, true));
    if (streets) {  // per street: instance table, leaf reach (not on the last street), one row of <= 4 root vectors per instance
        s->n_trunk_leaves = (int)st_leaf_ids.size();
        FAIL_IF(dev_upload(s, (const int32_t**)&s->d_trunk_leaves, st_leaf_ids));
        if (getenv("PRL_ST_DEBUG")) {
            fprintf(stderr, "street engine: %d streets, %d groups, %zu run-out chain roots (%d forest nodes), trunk leaves %d\n", s->st.n_levels, s->st.n_groups, s->st.chain.size(), s->chain_ft.n_nodes, s->n_trunk_leaves);
            for (int g = 0; g < s->st.n_groups; ++g)
                fprintf(stderr, "  group %d: street %d spec %d n_inst %d leaves %d cols/inst %d col_base %d last %d\n", g, s->st.group[g].street, s->st.group[g].spec, s->st.group[g].n_inst,
                        s->st.group[g].n_leaves, s->st.group[g].n_cols_inst, s->st.group[g].col_base, (int)s->st.group[g].last);
            for (int v = 0; v < PRL_ST_MAX_LEVELS; ++v) fprintf(stderr, "  street %d: %d rows, %d leaf slots\n", v, s->st.n_val_slots[v], s->st.n_leaf_slots[v]);
        }
        for (int g = 0; g < s->st.n_groups; ++g) {
            const PrlStLevelHost& H = s->st.group[g];
            if (H.street == 0) {
                // first street: an instance's root reach is read straight from the trunk's reach array (row = the trunk id of its chance leaf)
                std::vector<PrlStInst> inst0 = H.inst;
                for (PrlStInst& in : inst0) in.parent_slot = st_leaf_ids[in.parent_slot];
                FAIL_IF(dev_upload(s, (const PrlStInst**)&s->st_dev[g].inst, inst0));
            } else
            FAIL_IF(dev_upload(s, (const PrlStInst**)&s->st_dev[g].inst, H.inst));
        }
        for (int v = 0; v <= PRL_ST_MAX_LEVELS; ++v) {  // per street: the leaves' reach (both seats) and the rows of <= 4 root vectors, shared by its groups
            if (v < PRL_ST_MAX_LEVELS && s->st.n_leaf_slots[v] > 0) FAIL_IF(dev_alloc(s, &s->st_str[v].leaf_reach, (size_t)s->st.n_leaf_slots[v] * 2 * T.R));
            if (v < PRL_ST_MAX_LEVELS && s->st.n_val_slots[v] > 0) FAIL_IF(dev_alloc(s, &s->st_str[v].val, (size_t)s->st.n_val_slots[v] * 4 * T.R));
        }
        if (s->n_chain) {
            PrlDevTree& C = s->Tc;  // (plans, constants and chance weights were set above)
            const PrlFlatTree& u = s->chain_ft;
            C.n_nodes = u.n_nodes; C.n_cols = 0; C.n_levels = u.n_levels;
            FAIL_IF(dev_upload(s, &C.kind, u.kind));
            FAIL_IF(dev_upload(s, &C.actor, u.actor));
            FAIL_IF(dev_upload(s, &C.parent, u.parent));
            FAIL_IF(dev_upload(s, &C.child_idx, u.child_idx));
            FAIL_IF(dev_upload(s, &C.acted_last, u.acted_last));
            FAIL_IF(dev_upload(s, &C.board_id, u.board_id));
            FAIL_IF(dev_upload(s, &C.main_pot, u.main_pot));
            FAIL_IF(dev_upload(s, &C.n_children, u.n_children));
            FAIL_IF(dev_upload(s, &C.first_col, u.first_col));
            FAIL_IF(dev_upload(s, &C.child_start, u.child_start));
            FAIL_IF(dev_upload(s, &C.child_list, u.child_list));
            {   // the levels as the forest's own sum kernel walks them: chance nodes only (most roots are showdowns)
                std::vector<int32_t> ln;
                s->chain_level_start.assign(u.n_levels + 1, 0);
                for (int d = 0; d < u.n_levels; ++d) {
                    s->chain_level_start[d] = (int32_t)ln.size();
                    for (int k = u.level_start[d]; k < u.level_start[d + 1]; ++k)
                        if (u.kind[u.level_nodes[k]] == PRL_NODE_CHANCE) ln.push_back(u.level_nodes[k]);
                }
                s->chain_level_start[u.n_levels] = (int32_t)ln.size();
                if (ln.empty()) ln.push_back(0);
                FAIL_IF(dev_upload(s, &C.level_nodes, ln));
            }
            FAIL_IF(dev_alloc(s, &s->d_chain_ev, (size_t)u.n_nodes * 2 * T.R));
            PRL_HIP_TRY(hipMemsetAsync(s->d_chain_ev, 0, (size_t)u.n_nodes * 2 * T.R * sizeof(float), s->stream));
            std::vector<int32_t> node_kid(u.n_nodes, -1), root_of(u.n_nodes, -1), k_street, k_val;
            for (size_t i = 0; i < s->st.chain.size(); ++i) {
                node_kid[chain_root_id[i]] = (int32_t)i;
                k_street.push_back(s->st.chain[i].street);
                k_val.push_back(s->st.chain[i].val_slot);
            }
            std::vector<PrlStChainTerm> terms;
            for (int i = 0; i < u.n_nodes; ++i) {  // (parents come before their children: the forest keeps the flat tree's DFS order)
                root_of[i] = u.parent[i] < 0 ? node_kid[i] : root_of[u.parent[i]];
                if (u.kind[i] != PRL_NODE_TERM_SHOWDOWN) continue;
                const PrlStChainKid& ck = s->st.chain[root_of[i]];
                PrlStChainTerm tm = {};
                tm.node = i;
                tm.src_street = ck.street;
                tm.src_slot = ck.street == 0 ? st_leaf_ids[ck.parent_slot] : ck.parent_slot;  // (street 0: the trunk id of the all-in call's chance node)
                tm.kid = node_kid[i];
                std::vector<int> path;  // the chance nodes above it, bottom-up
                for (int a = u.parent[i]; a >= 0; a = u.parent[a]) path.push_back(a);
                FAIL_IF(path.size() + 1 > 3 ? (prl_set_error("street engine: a run-out chain deeper than three deals"), PRL_ERR_UNSUPPORTED) : PRL_OK);
                tm.w[tm.n_w++] = ck.w;
                for (size_t k = path.size(); k-- > 0;) tm.w[tm.n_w++] = chain_w[path[k]];
                terms.push_back(tm);
            }
            s->n_chain_term = (int)terms.size();
            // bundles: showdowns on one board share their workgroup's plan registers; short enough that the launch still fills the device
            std::stable_sort(terms.begin(), terms.end(), [&](const PrlStChainTerm& a, const PrlStChainTerm& b) { return u.board_id[a.node] < u.board_id[b.node]; });
            int per_bundle = (int)std::min<size_t>(8, std::max<size_t>(1, terms.size() / 8192));
            if (const char* e = getenv("PRL_ST_CHAIN_BUNDLE")) per_bundle = std::max(1, atoi(e));  // (tests: bundles on trees too small to form them)
            std::vector<PrlStChainBundle> bundles;
            for (size_t i = 0; i < terms.size();) {
                size_t j = i;
                while (j < terms.size() && j - i < (size_t)per_bundle && u.board_id[terms[j].node] == u.board_id[terms[i].node]) ++j;
                bundles.push_back(PrlStChainBundle{u.board_id[terms[i].node], (int32_t)i, (int32_t)(j - i)});
                i = j;
            }
            s->n_chain_bundles = (int)bundles.size();
            FAIL_IF(dev_upload(s, (const PrlStChainBundle**)&s->d_chain_bundles, bundles));
            FAIL_IF(dev_upload(s, (const PrlStChainTerm**)&s->d_chain_terms, terms));
            FAIL_IF(dev_upload(s, &s->chain_dev.street, k_street));
            FAIL_IF(dev_upload(s, &s->chain_dev.val_slot, k_val));
            FAIL_IF(dev_upload(s, &s->chain_dev.node_kid, node_kid));
            const char* cs = getenv("PRL_ST_CHAIN_STREAM");
            if (!cs || atoi(cs) != 0) {
                if (hipStreamCreate(&s->chain_stream) != hipSuccess || hipEventCreateWithFlags(&s->chain_fork, hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&s->chain_join, hipEventDisableTiming) != hipSuccess) {
                    prl_set_error("street engine: could not create the stream of the run-out forest");
                    prl_solver_destroy(s);
                    return PRL_ERR_HIP;
                }
            }
        }
#ifdef PRL_ST_TIMING
        FAIL_IF(dev_alloc(s, &s->sp.timing, (size_t)8));
        PRL_HIP_TRY(hipMemsetAsync(s->sp.timing, 0, 8 * sizeof(unsigned long long), s->stream));
#endif
    }
    if (fused) {
        // a row = what one first-deal outcome contributes to the chance sum: <= 4 root vectors per chance leaf of the trunk
        const size_t row_w = (size_t)(streets ? s->n_trunk_leaves : 1) * 4 * T.R;
        if (!streets) {
            // root-vector rows of a pass: one per board, or one per 32-board block when the pass sums its blocks itself (fused_board_pass)
            const bool rows_are_blocks = s->block_sum && (!exchange || shard_boards % PRL_CHANCE_BLOCK == 0);
            const size_t n_rows = rows_are_blocks ? ((size_t)full.n_boards + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK : (size_t)full.n_boards;
            FAIL_IF(dev_alloc(s, &s->d_board_out, n_rows * 4 * T.R));
        }
        FAIL_IF(dev_alloc(s, &s->d_row_sum, row_w));
        if (s->symmetrize) FAIL_IF(dev_alloc(s, &s->d_row_sym, row_w));
        const size_t n_blk = ((size_t)total_boards + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK + world;  // sized for the global board list
        const size_t n_grp = (n_blk + PRL_CHANCE_BLOCK - 1) / PRL_CHANCE_BLOCK;
        FAIL_IF(dev_alloc(s, &s->d_sum_scratch, (n_blk + n_grp + 1) * row_w));  // rows of up to 4 vectors (per trunk leaf)
        if (exchange) {
            // exchange whole canonical units: the highest summation level the shard size is a multiple of
            s->xlevel = shard_boards % (PRL_CHANCE_BLOCK * PRL_CHANCE_BLOCK) == 0 ? 2 : shard_boards % PRL_CHANCE_BLOCK == 0 ? 1 : 0;
            s->n_units = prl_fhp_units_at_level((int)shard_boards, s->xlevel);
            s->n_units_all = (world - 1) * s->n_units + prl_fhp_units_at_level((int)(total_boards - (int64_t)(world - 1) * shard_boards), s->xlevel);
            const size_t per_rank = (size_t)s->n_units * row_w;  // [units][<= 4 vectors (per trunk leaf)][R]
            FAIL_IF(dev_alloc(s, &s->d_xlocal, per_rank, true));
            FAIL_IF(dev_alloc(s, &s->d_xgather, per_rank * world, true));
            FAIL_IF(hipMemsetAsync(s->d_xlocal, 0, per_rank * sizeof(float), s->stream) == hipSuccess ? PRL_OK : PRL_ERR_HIP);  // a shorter last shard sends zero padding
        }
        FAIL_IF(dev_alloc(s, &s->d_half, (size_t)(streets ? s->n_trunk_leaves : 1) * 2 * T.R + 4));
        s->fp.regret = s->d_regret + s->board_ofs;
        s->fp.board_out = s->d_board_out;
#ifdef PRL_FHP_TIMING
        FAIL_IF(dev_alloc(s, &s->fp.timing, (size_t)96));
        PRL_HIP_TRY(hipMemsetAsync(s->fp.timing, 0, 96 * sizeof(unsigned long long), s->stream));
#endif
    }
    if (hipStreamSynchronize(s->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
        prl_set_error("device error while building the showdown plans");
        prl_solver_destroy(s);
        return PRL_ERR_HIP;
    }
#undef FAIL_IF
    *out = s;
    return prl_solver_reset(s);
}

int32_t prl_solver_create_sharded_ragged(const prl_tree_t* local_tree, int32_t variant, int32_t delay, int32_t world_size, int32_t rank,
                                         int64_t shard_boards, int64_t total_boards, prl_exchange_fn exchange, void* user, prl_solver_t** out) {
    if (world_size < 1 || rank < 0 || rank >= world_size || !exchange || shard_boards <= 0 || total_boards <= (int64_t)(world_size - 1) * shard_boards) {
        prl_set_error("bad world_size / rank / shard_boards / total_boards, or no exchange callback");
        return PRL_ERR_ARG;
    }
    return solver_create_impl(local_tree, variant, delay, PRL_ENGINE_FUSED, world_size, rank, exchange, user, out, shard_boards, total_boards);
}

// dummy non-null callback slot for the creation checks; replaced by prl_rccl_exchange once the communicator exists
static int32_t prl_exchange_placeholder(void*, const void*, void*, uint64_t) { return 1; }

int32_t prl_rccl_unique_id(void* out128) {
#if defined(PRL_EMU)
    (void)out128;
    prl_set_error("no RCCL in this build");
    return PRL_ERR_UNSUPPORTED;
#else
    if (!out128) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (!prl_rccl_api().ok) { prl_set_error("RCCL is not available: " + prl_rccl_api().why); return PRL_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    const ncclResult_t r = prl_rccl_api().GetUniqueId(&id);
    if (r != ncclSuccess) { prl_set_error(std::string("ncclGetUniqueId: ") + prl_rccl_api().GetErrorString(r)); return PRL_ERR_HIP; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, sizeof(id));
    return PRL_OK;
#endif
}

// the RCCL this library binds (the file its symbols come from), or why there is none: what to look at first when a multi-GPU run hangs
int32_t prl_rccl_info(char* out, int32_t n) {
    if (!out || n <= 0) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
#if defined(PRL_EMU)
    snprintf(out, (size_t)n, "no RCCL in this build");
    return PRL_ERR_UNSUPPORTED;
#else
    const PrlRcclApi& a = prl_rccl_api();
    snprintf(out, (size_t)n, "%s", a.ok ? a.path.c_str() : a.why.c_str());
    return a.ok ? PRL_OK : PRL_ERR_UNSUPPORTED;
#endif
}

int32_t prl_solver_create_sharded_rccl(const prl_tree_t* local_tree, int32_t variant, int32_t delay, int32_t world_size, int32_t rank,
                                       const void* unique_id128, int64_t shard_boards, int64_t total_boards, prl_solver_t** out) {
#if defined(PRL_EMU)
    prl_set_error("no RCCL in this build");
    return PRL_ERR_UNSUPPORTED;
#else
    if (world_size < 1 || rank < 0 || rank >= world_size || !unique_id128) { prl_set_error("bad world_size / rank / unique id"); return PRL_ERR_ARG; }
    if (shard_boards > 0 && total_boards <= (int64_t)(world_size - 1) * shard_boards) { prl_set_error("bad shard_boards / total_boards"); return PRL_ERR_ARG; }
    if (!prl_rccl_api().ok) { prl_set_error("RCCL is not available: " + prl_rccl_api().why); return PRL_ERR_UNSUPPORTED; }
    return solver_create_impl(local_tree, variant, delay, PRL_ENGINE_FUSED, world_size, rank, prl_exchange_placeholder, nullptr, out,
                              shard_boards > 0 ? shard_boards : 0, shard_boards > 0 ? total_boards : 0, unique_id128);
#endif
}

int32_t prl_solver_create_opts(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t flags, prl_solver_t** out) {
    if (flags & ~PRL_SOLVER_AVG_F32) { prl_set_error("unknown solver flag"); return PRL_ERR_ARG; }
    return solver_create_impl(tree, variant, delay, engine, 1, 0, nullptr, nullptr, out, 0, 0, nullptr, flags);
}

int32_t prl_solver_create_weighted(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t flags, const int32_t* board_mult, int32_t symmetrize,
                                   prl_solver_t** out) {
    if (flags & ~PRL_SOLVER_AVG_F32) { prl_set_error("unknown solver flag"); return PRL_ERR_ARG; }
    if (!board_mult) { prl_set_error("board_mult is NULL"); return PRL_ERR_ARG; }
    return solver_create_impl(tree, variant, delay, PRL_ENGINE_FUSED, 1, 0, nullptr, nullptr, out, 0, 0, nullptr, flags, board_mult, symmetrize);
}

static int32_t create_placed_impl(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t flags, int32_t n_candidates,
                                  int32_t probe_iters, float* out_ms, int32_t* out_chosen, prl_solver_t** out, const int32_t* board_mult, int32_t symmetrize);

int32_t prl_solver_create_placed(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t flags, int32_t n_candidates,
                                 int32_t probe_iters, float* out_ms, int32_t* out_chosen, prl_solver_t** out) {
    return create_placed_impl(tree, variant, delay, engine, flags, n_candidates, probe_iters, out_ms, out_chosen, out, nullptr, 0);
}

// placement selection for weighted boards / suit classes (the whole game is 30 GB: three candidates fit beside one another)
int32_t prl_solver_create_weighted_placed(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t flags, const int32_t* board_mult, int32_t symmetrize,
                                          int32_t n_candidates, int32_t probe_iters, float* out_ms, int32_t* out_chosen, prl_solver_t** out) {
    if (!board_mult) { prl_set_error("prl_solver_create_weighted_placed: board multiplicities"); return PRL_ERR_ARG; }
    return create_placed_impl(tree, variant, delay, PRL_ENGINE_FUSED, flags, n_candidates, probe_iters, out_ms, out_chosen, out, board_mult, symmetrize);
}

static int32_t create_placed_impl(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t flags, int32_t n_candidates,
                                  int32_t probe_iters, float* out_ms, int32_t* out_chosen, prl_solver_t** out, const int32_t* board_mult, int32_t symmetrize) {
    if (!out || n_candidates < 1 || n_candidates > 8 || probe_iters < 1) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (out_ms) for (int i = 0; i < n_candidates; ++i) out_ms[i] = 0.f;
    if (out_chosen) *out_chosen = 0;
    std::vector<prl_solver*> cand;
    std::vector<float> ms;
    int rc = PRL_OK;
    for (int i = 0; i < n_candidates; ++i) {
        prl_solver* s = nullptr;
        rc = board_mult ? prl_solver_create_weighted(tree, variant, delay, flags, board_mult, symmetrize, &s) : prl_solver_create_opts(tree, variant, delay, engine, flags, &s);
        if (rc != PRL_OK) {
            if (rc == PRL_ERR_OOM && !cand.empty()) {  // no room for another set of arrays: choose among those there are
                (void)hipGetLastError();  // the tolerated hipMalloc failure must not be what the survivor's next PRL_HIP_TRY(hipGetLastError()) reports
                rc = PRL_OK;
                break;
            }
            for (prl_solver* c : cand) prl_solver_destroy(c);
            return rc;
        }
        cand.push_back(s);
        if (!s->fused) break;  // nothing to choose
        // past the first iterations (uniform strategies, first averages): the steady-state passes are what is compared
        float t = 0.f;
        rc = prl_solver_iterations(s, 3);
        if (rc == PRL_OK) rc = prl_solver_sync(s);
        if (rc == PRL_OK) rc = prl_solver_time_iterations(s, probe_iters, &t);
        if (rc != PRL_OK) { for (prl_solver* c : cand) prl_solver_destroy(c); return rc; }
        ms.push_back(t / (float)probe_iters);
    }
    size_t best = 0;
    for (size_t i = 1; i < ms.size(); ++i) if (ms[i] < ms[best]) best = i;
    for (size_t i = 0; i < cand.size(); ++i) if (i != best) prl_solver_destroy(cand[i]);
    if (out_ms) for (size_t i = 0; i < ms.size(); ++i) out_ms[i] = ms[i];
    if (out_chosen) *out_chosen = (int32_t)best;
    if (!ms.empty()) {
        rc = prl_solver_reset(cand[best]);
        if (rc != PRL_OK) { prl_solver_destroy(cand[best]); *out = nullptr; return rc; }  // the caller gets no handle, so nothing may stay allocated
    }
    *out = cand[best];
    return PRL_OK;
}

int32_t prl_solver_create_ex(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, prl_solver_t** out) {
    return solver_create_impl(tree, variant, delay, engine, 1, 0, nullptr, nullptr, out);
}

int32_t prl_solver_create_sharded(const prl_tree_t* local_tree, int32_t variant, int32_t delay, int32_t world_size, int32_t rank,
                                  prl_exchange_fn exchange, void* user, prl_solver_t** out) {
    if (world_size < 1 || rank < 0 || rank >= world_size) { prl_set_error("bad world_size / rank"); return PRL_ERR_ARG; }
    if (world_size > 1 && !exchange) { prl_set_error("sharded solve needs an exchange callback"); return PRL_ERR_ARG; }
    // with a callback the exchange path is taken even for world_size 1 (a one-rank all-gather): same code on any world size
    return solver_create_impl(local_tree, variant, delay, exchange ? PRL_ENGINE_FUSED : PRL_ENGINE_AUTO, world_size, rank, exchange, user, out);
}

namespace {
struct PrlStateHeader {
    uint32_t magic, version;
    int32_t variant, delay, fused, iter, full_cols, R, trunk_cols, trunk_nodes, src0, src1, board_avg_f64, has_avg_sum;
    uint64_t fingerprint;  // prl_solver::fingerprint of the saving solver
};
const uint32_t PRL_STATE_VERSION = 3;  // 3: single-deal fused solvers save their column arrays as they hold them (sorted storage)
const uint32_t PRL_STATE_MAGIC = 0x50524C53u;  // "PRLS"

struct StateLayout { size_t regret, avg, avg_sum, strategy, strat_f64, avg_f64, hist, total; };
StateLayout state_layout(const prl_solver* s, int iter) {
    const size_t nc = col_array_elems(s), tc = (size_t)s->T.n_cols * s->R;
    StateLayout L;
    size_t o = sizeof(PrlStateHeader);
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 15) & ~(size_t)15; return at; };
    L.regret = take(nc * 4); L.avg = take(nc * 8); L.avg_sum = take(s->S.avg_sum ? nc * 4 : 0); L.strategy = take(tc * 8);
    L.strat_f64 = take((size_t)s->T.n_nodes); L.avg_f64 = take((size_t)s->T.n_nodes); L.hist = take((size_t)(iter + 1) * 2 * 4);
    L.total = o;
    return L;
}
}  // namespace

int32_t prl_solver_state_size(prl_solver_t* s, uint64_t* out) {
    if (!s || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (s->avg_f32) { prl_set_error("checkpoints: not for solvers with the float32 running average (PRL_SOLVER_AVG_F32)"); return PRL_ERR_UNSUPPORTED; }
    *out = (uint64_t)state_layout(s, s->iter).total;
    return PRL_OK;
}

int32_t prl_solver_save_state(prl_solver_t* s, void* out, uint64_t bytes) {
    if (!s || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (s->avg_f32) { prl_set_error("save_state: not for solvers with the float32 running average (PRL_SOLVER_AVG_F32)"); return PRL_ERR_UNSUPPORTED; }
    if (s->user_strategy_f64 >= 0) { prl_set_error("save_state: an explicit strategy is loaded (set_strategy); reset or fill_uniform first"); return PRL_ERR_STATE; }
    TRY(ensure_ev(s));  // closes a pending evaluation / pending average updates, so the blob is a clean iteration boundary
    const StateLayout L = state_layout(s, s->iter);
    if (bytes < L.total) { prl_set_error("save_state: buffer too small"); return PRL_ERR_ARG; }
    if (s->avg_pending[0] >= 0 || s->avg_pending[1] >= 0 || s->expl_pending) { prl_set_error("save_state: iteration not closed"); return PRL_ERR_STATE; }
    TRY(ensure_board_avg(s));
    PrlStateHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = PRL_STATE_MAGIC; h.version = PRL_STATE_VERSION; h.fingerprint = s->fingerprint; h.variant = s->variant; h.delay = s->delay; h.fused = s->fused; h.iter = s->iter;
    h.full_cols = s->full_cols; h.R = s->R; h.trunk_cols = s->T.n_cols; h.trunk_nodes = s->T.n_nodes; h.src0 = s->src[0]; h.src1 = s->src[1];
    h.board_avg_f64 = s->board_avg_f64; h.has_avg_sum = s->S.avg_sum != nullptr;
    char* b = (char*)out;
    memcpy(b, &h, sizeof(h));
    const size_t nc = col_array_elems(s), tc = (size_t)s->T.n_cols * s->R;
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    PRL_HIP_TRY(hipMemcpy(b + L.regret, s->d_regret, nc * 4, hipMemcpyDeviceToHost));
    PRL_HIP_TRY(hipMemcpy(b + L.avg, s->d_avg, nc * 8, hipMemcpyDeviceToHost));
    if (s->S.avg_sum) PRL_HIP_TRY(hipMemcpy(b + L.avg_sum, s->S.avg_sum, nc * 4, hipMemcpyDeviceToHost));
    PRL_HIP_TRY(hipMemcpy(b + L.strategy, s->S.strategy, tc * 8, hipMemcpyDeviceToHost));
    PRL_HIP_TRY(hipMemcpy(b + L.strat_f64, s->S.strat_f64, (size_t)s->T.n_nodes, hipMemcpyDeviceToHost));
    PRL_HIP_TRY(hipMemcpy(b + L.avg_f64, s->S.avg_f64, (size_t)s->T.n_nodes, hipMemcpyDeviceToHost));
    PRL_HIP_TRY(hipMemcpy(b + L.hist, s->d_expl_hist, (size_t)(s->iter + 1) * 2 * 4, hipMemcpyDeviceToHost));
    return PRL_OK;
}

int32_t prl_solver_load_state(prl_solver_t* s, const void* in, uint64_t bytes) {
    if (!s || !in || bytes < sizeof(PrlStateHeader)) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PrlStateHeader h;
    memcpy(&h, in, sizeof(h));
    if (s->avg_f32) { prl_set_error("load_state: not for solvers with the float32 running average (PRL_SOLVER_AVG_F32)"); return PRL_ERR_UNSUPPORTED; }
    if (h.magic != PRL_STATE_MAGIC) { prl_set_error("load_state: not a solver state blob"); return PRL_ERR_ARG; }
    if (h.version != PRL_STATE_VERSION) {  // version 3 (round 4) stores the sorted board storage as it is held: older blobs cannot be read ("Checkpoints")
        prl_set_error("load_state: state blob of format version " + std::to_string(h.version) + ", this library reads version " + std::to_string(PRL_STATE_VERSION) +
                      " only (no migration: re-solve, or load with the library that wrote it and hand the columns over through prl_solver_get / prl_solver_set_strategy)");
        return PRL_ERR_ARG;
    }
    if (h.variant != s->variant || h.delay != s->delay || h.fused != (int32_t)s->fused || h.full_cols != s->full_cols || h.R != s->R ||
        h.trunk_cols != s->T.n_cols || h.trunk_nodes != s->T.n_nodes || h.has_avg_sum != (int32_t)(s->S.avg_sum != nullptr) || h.iter < 0) {
        prl_set_error("load_state: the blob was saved by a solver with a different tree / variant / delay / engine");
        return PRL_ERR_STATE;
    }
    if (h.fingerprint != s->fingerprint) {
        prl_set_error("load_state: the blob belongs to another board list / game / stack sizes or to another rank's shard");
        return PRL_ERR_STATE;
    }
    if (h.src0 < 0 || h.src0 > PRL_SRC_ARR32 || h.src1 < 0 || h.src1 > PRL_SRC_ARR32) { prl_set_error("load_state: corrupt header"); return PRL_ERR_ARG; }
    const StateLayout L = state_layout(s, h.iter);
    if (bytes < L.total) { prl_set_error("load_state: truncated blob"); return PRL_ERR_ARG; }
    const char* b = (const char*)in;
    const size_t nc = col_array_elems(s), tc = (size_t)s->T.n_cols * s->R;
    TRY(ensure_hist(s, h.iter + 1));
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    PRL_HIP_TRY(hipMemcpy(s->d_regret, b + L.regret, nc * 4, hipMemcpyHostToDevice));
    PRL_HIP_TRY(hipMemcpy(s->d_avg, b + L.avg, nc * 8, hipMemcpyHostToDevice));
    if (s->S.avg_sum) PRL_HIP_TRY(hipMemcpy(s->S.avg_sum, b + L.avg_sum, nc * 4, hipMemcpyHostToDevice));
    PRL_HIP_TRY(hipMemcpy(s->S.strategy, b + L.strategy, tc * 8, hipMemcpyHostToDevice));
    PRL_HIP_TRY(hipMemcpy(s->S.strat_f64, b + L.strat_f64, (size_t)s->T.n_nodes, hipMemcpyHostToDevice));
    PRL_HIP_TRY(hipMemcpy(s->S.avg_f64, b + L.avg_f64, (size_t)s->T.n_nodes, hipMemcpyHostToDevice));
    PRL_HIP_TRY(hipMemcpy(s->d_expl_hist, b + L.hist, (size_t)(h.iter + 1) * 2 * 4, hipMemcpyHostToDevice));
    s->iter = h.iter; s->src[0] = h.src0; s->src[1] = h.src1; s->board_avg_f64 = h.board_avg_f64 != 0; s->board_avg_stale = false;
    s->user_strategy_f64 = -1; s->expl_pending = false; s->have_half = false; s->avg_pending[0] = s->avg_pending[1] = -1;
    s->ev_valid = false;
    for (double& a : s->blk_avg) a = 0.;  // the blocked hands' average: a function of the iteration count alone
    for (int it = 0; it < s->iter; ++it) {
        int mode; double m_old, m_new;
        cfr_plus_weights(s, it, &mode, &m_old, &m_new);
        blocked_avg_step(s, mode, m_old, m_new);
    }
    return do_update_reach(s, s->S);
}

int32_t prl_solver_get_stream(prl_solver_t* s, void** out) {
    if (!s || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    *out = (void*)s->stream;
    return PRL_OK;
}

int32_t prl_solver_set_exchange_async(prl_solver_t* s, int32_t on) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    s->exchange_async = on != 0;
    return PRL_OK;
}

int32_t prl_chance_sum_host_ragged(const float* board_values, int32_t n_boards, int32_t R, int32_t world, int32_t shard_boards, float* out) {
    if (!board_values || !out || n_boards <= 0 || R <= 0 || world < 1 || shard_boards <= 0 || (int64_t)(world - 1) * shard_boards >= n_boards ||
        (int64_t)world * shard_boards < n_boards) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device"); return PRL_ERR_NO_DEVICE; }
    const size_t R2 = (size_t)2 * R, nv = (size_t)n_boards * R2;
    const int n_last = n_boards - (world - 1) * shard_boards;
    const int level = world == 1 ? 0 : shard_boards % (PRL_CHANCE_BLOCK * PRL_CHANCE_BLOCK) == 0 ? 2 : shard_boards % PRL_CHANCE_BLOCK == 0 ? 1 : 0;
    const int n_units = prl_fhp_units_at_level(shard_boards, level);
    const int n_units_all = (world - 1) * n_units + prl_fhp_units_at_level(n_last, level);
    float *d_vals = nullptr, *d_scratch = nullptr, *d_gather = nullptr, *d_compact = nullptr, *d_out = nullptr;
    int rc = PRL_OK;
#define CS_TRY(x) do { if ((x) != hipSuccess) { prl_set_error("HIP error in prl_chance_sum_host"); rc = PRL_ERR_HIP; goto done; } } while (0)
    CS_TRY(hipMalloc((void**)&d_vals, nv * sizeof(float)));
    CS_TRY(hipMalloc((void**)&d_scratch, ((size_t)n_boards / PRL_CHANCE_BLOCK + n_boards / (PRL_CHANCE_BLOCK * PRL_CHANCE_BLOCK) + 4 + world) * R2 * sizeof(float)));
    CS_TRY(hipMalloc((void**)&d_gather, (size_t)world * n_units * R2 * sizeof(float)));
    CS_TRY(hipMalloc((void**)&d_compact, (size_t)world * n_units * R2 * sizeof(float)));
    CS_TRY(hipMalloc((void**)&d_out, R2 * sizeof(float)));
    CS_TRY(hipMemcpy(d_vals, board_values, nv * sizeof(float), hipMemcpyHostToDevice));
    CS_TRY(hipMemset(d_gather, 0, (size_t)world * n_units * R2 * sizeof(float)));
    if (world == 1) prl_launch_fhp_chance_sum(d_vals, n_boards, 2 * R, d_scratch, d_out, nullptr);
    else {
        for (int r = 0; r < world; ++r)  // rank r's units land where the all-gather would put them
            prl_launch_fhp_chance_partial(d_vals + (size_t)r * shard_boards * R2, r == world - 1 ? n_last : shard_boards, level, 2 * R, d_scratch,
                                          d_gather + (size_t)r * n_units * R2, nullptr);
        prl_launch_fhp_compact_gathered(d_gather, world, 1, n_units, 2 * R, d_compact, nullptr);
        prl_launch_fhp_chance_finish(d_compact, n_units_all, level, 2 * R, d_scratch, d_out, nullptr);
    }
    CS_TRY(hipDeviceSynchronize());
    CS_TRY(hipMemcpy(out, d_out, R2 * sizeof(float), hipMemcpyDeviceToHost));
#undef CS_TRY
done:
    (void)hipFree(d_vals); (void)hipFree(d_scratch); (void)hipFree(d_gather); (void)hipFree(d_compact); (void)hipFree(d_out);
    return rc;
}

int32_t prl_chance_sum_host(const float* board_values, int32_t n_boards, int32_t R, int32_t world, float* out) {
    if (world < 1 || n_boards <= 0 || n_boards % world) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    return prl_chance_sum_host_ragged(board_values, n_boards, R, world, n_boards / world, out);
}

int32_t prl_solver_create(const prl_tree_t* tree, int32_t variant, int32_t delay, prl_solver_t** out) {
    return prl_solver_create_ex(tree, variant, delay, PRL_ENGINE_AUTO, out);
}

void prl_solver_destroy(prl_solver_t* s) {
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
#if !defined(PRL_EMU)
    if (s->levels_graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)s->levels_graph_exec);
#endif
#if !defined(PRL_EMU)
    if (s->rccl_comm) (void)prl_rccl_api().CommDestroy((ncclComm_t)s->rccl_comm);
#endif
    for (void* p : s->allocs) (void)hipFree(p);
#if !defined(PRL_EMU)
    for (void* r : s->vmm) { vmm_free(*(PrlVmmRange*)r); delete (PrlVmmRange*)r; }
#endif
    if (s->chain_fork) (void)hipEventDestroy(s->chain_fork);
    if (s->chain_join) (void)hipEventDestroy(s->chain_join);
    if (s->chain_stream) (void)hipStreamDestroy(s->chain_stream);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

// CFRBase.reset (_CFRBase.py:110-120): clear regrets / averages, uniform strategy, reach, EV (+ exploitability)
int32_t prl_solver_reset(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    const size_t nc = col_array_elems(s);
    s->iter = 0;
    for (double& a : s->blk_avg) a = 0.;
    s->expl_pending = false;
    s->have_half = false;
    s->avg_pending[0] = s->avg_pending[1] = -1;
    PRL_HIP_TRY(hipMemsetAsync(s->d_regret, 0, nc * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(s->d_avg, 0, (s->avg_f32 ? (size_t)s->T.n_cols * s->R : nc) * sizeof(double), s->stream));
    if (s->avg_f32) PRL_HIP_TRY(hipMemsetAsync(s->d_avg32, 0, nc * sizeof(float), s->stream));
    PRL_HIP_TRY(hipMemsetAsync(s->S.avg_f64, 0, (size_t)s->T.n_nodes, s->stream));
    if (s->S.avg_sum) PRL_HIP_TRY(hipMemsetAsync(s->S.avg_sum, 0, nc * sizeof(float), s->stream));
    s->board_avg_f64 = false; s->board_avg_stale = false;
    TRY(prl_solver_fill_uniform(s));
    TRY(ensure_ev(s));
    return record_expl(s);
}

int32_t prl_solver_fill_uniform(prl_solver_t* s) {  // PublicTree.fill_uniform_random (StrategyFiller.py:17-24)
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipMemsetAsync(s->S.strat_f64, 0, (size_t)s->T.n_nodes, s->stream));
    prl_launch_fill_uniform(s->T, s->S, s->d_col_node, s->stream);
    s->src[0] = s->src[1] = PRL_SRC_UNIFORM64;
    s->user_strategy_f64 = -1;
    s->have_half = false;
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

// arbitrary strategy, column-major [n_cols][R] (= node.strategy.T per decision node, DFS order); float32 or float64 host data.
// Equivalent of fill_with_agent_policy / fill_random_random + update_reach_probs (StrategyFiller.py:26-43).
int32_t prl_solver_set_strategy(prl_solver_t* s, const void* strat, int32_t is_f64) {
    return prl_solver_set_strategy_mixed(s, strat, is_f64, nullptr);
}

// node_is_f64 (may be NULL = every node like the array): per NODE, whether its strategy is a float64 array in the reference's tree
// (uniform fills and averages are float64, regret-matched strategies float32 -- the arithmetic dtype follows it, SURVEY 8a dtype
// ledger); needs float64 host data. What lets Python-side CFR variants (CFRBase's hook methods) mix both as the reference does.
int32_t prl_solver_set_strategy_mixed(prl_solver_t* s, const void* strat, int32_t is_f64, const uint8_t* node_is_f64) {
    if (!s || !strat) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (node_is_f64 && (!is_f64 || s->fused)) { prl_set_error("per-node strategy dtypes: float64 data, LEVELS engine"); return PRL_ERR_ARG; }
    const size_t nc = (size_t)s->full_cols * s->R;
    std::vector<double> tmp;
    std::vector<char> perm;  // the caller's columns (flat-tree DFS order) in the engine's internal column order
    if (!s->col_dfs.empty()) {
        const size_t cb = (size_t)s->R * (is_f64 ? sizeof(double) : sizeof(float));
        perm.resize((size_t)s->full_cols * cb);
        for (int c = 0; c < s->full_cols; ++c) memcpy(perm.data() + (size_t)c * cb, (const char*)strat + (size_t)s->col_dfs[c] * cb, cb);
        strat = perm.data();
    }
    const double* src = (const double*)strat;
    if (!is_f64) {
        const size_t nconv = s->fused ? (size_t)s->T.n_cols * s->R : nc;  // (FUSED: only the trunk's columns are kept as float64)
        tmp.resize(nconv);
        const float* f = (const float*)strat;
        for (size_t i = 0; i < nconv; ++i) tmp[i] = (double)f[i];  // exact widening: storage only, arithmetic stays float32
        src = tmp.data();
    }
    if (s->sorted) {
        // the trunk's columns as they are, the boards' columns into sorted storage (what the caller had for the hands a board blocks is
        // kept aside so that prl_solver_get returns it)
        const size_t ne = col_array_elems(s), tc = (size_t)s->T.n_cols * s->R, nblk = (size_t)s->fp.n_boards * s->ncb * PRL_FHP_NBLOCKED;
        if (is_f64) {
            if (!s->d_user_strategy) TRY(dev_alloc(s, &s->d_user_strategy, ne));
            if (!s->d_user_blocked64) TRY(dev_alloc(s, &s->d_user_blocked64, nblk));
            PRL_HIP_TRY(hipMemcpyAsync(s->d_user_strategy, strat, tc * sizeof(double), hipMemcpyHostToDevice, s->stream));
            TRY(sorted_set_boards(s, (const double*)strat + tc, 8, s->d_user_strategy + s->board_ofs, s->d_user_blocked64));
        } else {
            if (!s->d_user_strategy32) TRY(dev_alloc(s, &s->d_user_strategy32, ne));
            if (!s->d_user_blocked32) TRY(dev_alloc(s, &s->d_user_blocked32, nblk));
            PRL_HIP_TRY(hipMemcpyAsync(s->d_user_strategy32, strat, tc * sizeof(float), hipMemcpyHostToDevice, s->stream));
            TRY(sorted_set_boards(s, (const float*)strat + tc, 4, s->d_user_strategy32 + s->board_ofs, s->d_user_blocked32));
        }
        s->user_strategy_f64 = is_f64 ? 1 : 0;
    } else if (s->fused) {
        if (is_f64) {
            if (!s->d_user_strategy) TRY(dev_alloc(s, &s->d_user_strategy, nc));
            PRL_HIP_TRY(hipMemcpyAsync(s->d_user_strategy, src, nc * sizeof(double), hipMemcpyHostToDevice, s->stream));
        } else {
            if (!s->d_user_strategy32) TRY(dev_alloc(s, &s->d_user_strategy32, nc));
            PRL_HIP_TRY(hipMemcpyAsync(s->d_user_strategy32, strat, nc * sizeof(float), hipMemcpyHostToDevice, s->stream));
        }
        s->user_strategy_f64 = is_f64 ? 1 : 0;
    }
    // LEVELS: the whole array; FUSED: the trunk columns (they precede the board columns)
    PRL_HIP_TRY(hipMemcpyAsync(s->S.strategy, src, (size_t)s->T.n_cols * s->R * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (node_is_f64) PRL_HIP_TRY(hipMemcpyAsync(s->S.strat_f64, node_is_f64, (size_t)s->T.n_nodes, hipMemcpyHostToDevice, s->stream));
    else PRL_HIP_TRY(hipMemsetAsync(s->S.strat_f64, is_f64 ? 1 : 0, (size_t)s->T.n_nodes, s->stream));
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));  // tmp / the caller's buffers may go away
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

int32_t prl_solver_update_reach(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

int32_t prl_solver_compute_ev(prl_solver_t* s) {  // PublicTree.compute_ev (PublicTree.py:128)
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    s->ev_valid = false;
    return ensure_ev(s);
}

// One CFRBase.iteration() (_CFRBase.py:122-134) without _evaluate_avg_strats (see prl_solver_eval_avg). Asynchronous.
// LEVELS: the reference recomputes the EVs at the top of the p = 0 half although nothing changed since the pass that
// closed the previous iteration; that pass is reused (identical values), so an iteration costs two EV passes, not three.
// FUSED: every half-iteration is one board pass that computes the EVs and updates that seat's regrets in place, then the
// trunk is updated with the summed chance-node values; a third (best-response) pass yields the exploitability.
static int iteration_core(prl_solver* s, bool closing_eval) {
    if (s->fused && s->user_strategy_f64 >= 0) { prl_set_error("call reset() / fill_uniform() before iterating after set_strategy()"); return PRL_ERR_STATE; }
    int mode = 0;
    double m_old = 0., m_new = 0.;
    cfr_plus_weights(s, s->iter, &mode, &m_old, &m_new);
    s->fp.avg_mode = mode;
    s->fp.m_old = m_old;
    s->fp.m_new = m_new;
    for (int p = 0; p < 2; ++p) {
        bool second_half = false;
        if (s->fused) {
            const bool steady = s->src[0] == PRL_SRC_REGRET && s->src[1] == PRL_SRC_REGRET;
            if (p == 0 && s->expl_pending && s->have_half) {
                // seat 0's half of the previous iterate's exploitability rides on this pass; seat 1's half was computed
                // by the pass that updated seat 1
                TRY(expl_to_history(s));
                TRY(do_compute_ev(s, s->S, PRL_FHP_UPDATE0_BR));
                s->expl_pending = false;
                s->have_half = false;
            } else if (p == 0 && s->expl_pending) {  // the evaluation that closes the previous iteration rides on this pass
                TRY(expl_to_history(s));
                TRY(do_compute_ev(s, s->S, PRL_FHP_UPDATE0_EVAL));
                s->expl_pending = false;
            } else if (p == 1 && steady) {
                TRY(do_compute_ev(s, s->S, PRL_FHP_UPDATE1_EVAL1));
                second_half = true;
            } else TRY(do_compute_ev(s, s->S, p == 0 ? PRL_FHP_UPDATE0 : PRL_FHP_UPDATE1));
        } else TRY(ensure_ev(s));
        prl_launch_regret_strategy(s->T, s->S, s->d_nodes_p[p], s->n_nodes_p[p], p, s->variant, s->iter, s->stream);
        s->src[p] = PRL_SRC_REGRET;
        s->ev_valid = false;
        TRY(do_update_reach(s, s->S));
        prl_launch_average(s->T, s->S, s->d_nodes_p[p], s->n_nodes_p[p], p, s->variant, s->iter, mode, m_old, m_new, s->stream);
        if (s->fused && mode) s->board_avg_f64 = mode == 2;  // the board columns were averaged inside the board pass
        if (s->fused && s->variant != PRL_CFR_PLUS) s->avg_pending[p] = s->iter;  // applied by the next pass that walks seat p
        if (second_half) s->have_half = true;  // d_half: seat 1's value / best response under the updated strategies
    }
    s->fp.avg_mode = 0;
    if (s->sorted) blocked_avg_step(s, mode, m_old, m_new);
    s->iter += 1;
    if (s->fused && !closing_eval) {
        s->expl_pending = true;
        PRL_HIP_TRY(hipGetLastError());
        return PRL_OK;
    }
    if (s->fused && s->have_half) {  // seat 1's half is known: seat 0's batch with best response closes the iteration
        TRY(do_compute_ev(s, s->S, PRL_FHP_EVAL0));
        s->have_half = false;
        s->ev_valid = true;
    } else TRY(ensure_ev(s));
    PRL_HIP_TRY(hipGetLastError());
    return record_expl(s);
}

// LEVELS engine, small 1-hole-card trees: whole iterations inside one single-workgroup kernel (prl_k_small_iterations)
static int small_tree_iterations(prl_solver* s, int n) {
    TRY(ensure_ev(s));  // an iteration starts from the evaluation that closed the previous one
    TRY(ensure_hist(s, s->iter + n + 1));
    if (!s->d_ip) TRY(dev_alloc(s, &s->d_ip, (size_t)1));
    if (!s->d_level_start) TRY(dev_upload(s, &s->d_level_start, s->ft.level_start));
    PrlIterDev ip;
    memset(&ip, 0, sizeof(ip));
    ip.iter = s->iter;
    ip.hist = s->d_expl_hist;
    PRL_HIP_TRY(hipMemcpyAsync(s->d_ip, &ip, sizeof(ip), hipMemcpyHostToDevice, s->stream));
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));  // `ip` is a stack variable
    for (int done = 0; done < n;) {
        const int k = n - done < 256 ? n - done : 256;  // bounded kernel run time
        prl_launch_small_iterations(s->T, s->S, s->d_level_start, s->d_term_nodes, s->n_term, s->d_nodes_p[0], s->n_nodes_p[0], s->d_nodes_p[1],
                                    s->n_nodes_p[1], s->variant, s->delay, k, s->d_ip, s->stream);
        done += k;
    }
    s->iter += n;
    s->src[0] = s->src[1] = PRL_SRC_REGRET;
    s->ev_valid = true;
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

// Many independent small-tree solves in ONE launch, one workgroup (CU) per solver: a Leduc-sized tree occupies a single CU, so
// a sweep over games / stack sizes / bet sets fills the GPU only this way. Every solver advances n iterations exactly as
// prl_solver_iterations(s, n) would (same kernel body, same bits).
extern "C" int32_t prl_solver_iterations_many(prl_solver_t** solvers, int32_t n_solvers, int32_t n) {
    if (!solvers || n_solvers <= 0 || n_solvers > 65535 || n < 0) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    for (int i = 0; i < n_solvers; ++i) {
        prl_solver* s = solvers[i];
        if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
        if (s->fused || !s->small_tree) { prl_set_error("iterations_many: every solver must be a small 1-hole-card tree on the LEVELS engine"); return PRL_ERR_UNSUPPORTED; }
        for (int j = 0; j < i; ++j) if (solvers[j] == s) { prl_set_error("iterations_many: duplicate solver"); return PRL_ERR_ARG; }
    }
    if (n == 0) return PRL_OK;
    std::vector<PrlSmallJob> jobs((size_t)n_solvers);
    size_t lds_max = 0;
    for (int i = 0; i < n_solvers; ++i) {
        prl_solver* s = solvers[i];
        TRY(ensure_ev(s));  // an iteration starts from the evaluation that closed the previous one
        TRY(ensure_hist(s, s->iter + n + 1));
        if (!s->d_ip) TRY(dev_alloc(s, &s->d_ip, (size_t)1));
        if (!s->d_level_start) TRY(dev_upload(s, &s->d_level_start, s->ft.level_start));
        PrlIterDev ip;
        memset(&ip, 0, sizeof(ip));
        ip.iter = s->iter;
        ip.hist = s->d_expl_hist;
        PRL_HIP_TRY(hipMemcpyAsync(s->d_ip, &ip, sizeof(ip), hipMemcpyHostToDevice, s->stream));
        PRL_HIP_TRY(hipStreamSynchronize(s->stream));  // `ip` is a stack variable; the solver's earlier work is complete
        size_t lds = 0;
        bool in_lds = false, tree_lds = false;
        prl_small_lds_plan(s->T, s->S, s->n_term, s->n_nodes_p[0], s->n_nodes_p[1], &in_lds, &tree_lds, &lds);
        if (lds > lds_max) lds_max = lds;
        PrlSmallJob& J = jobs[(size_t)i];
        memset(&J, 0, sizeof(J));
        J.T = s->T; J.S = s->S; J.level_start = s->d_level_start; J.term_nodes = s->d_term_nodes; J.n_term = s->n_term;
        J.nodes_p[0] = s->d_nodes_p[0]; J.nodes_p[1] = s->d_nodes_p[1]; J.n_nodes_p[0] = s->n_nodes_p[0]; J.n_nodes_p[1] = s->n_nodes_p[1];
        J.variant = s->variant; J.delay = s->delay; J.state_in_lds = in_lds ? 1 : 0; J.tree_in_lds = tree_lds ? 1 : 0; J.n_cols = s->T.n_cols; J.ip = s->d_ip;
    }
    prl_solver* s0 = solvers[0];
    PrlSmallJob* d_jobs = nullptr;
    PRL_HIP_TRY(hipMalloc((void**)&d_jobs, jobs.size() * sizeof(PrlSmallJob)));
    int rc = PRL_OK;
    for (int done = 0; done < n && rc == PRL_OK;) {
        const int k = n - done < 256 ? n - done : 256;  // bounded kernel run time
        for (auto& J : jobs) J.n_iters = k;
        if (hipMemcpy(d_jobs, jobs.data(), jobs.size() * sizeof(PrlSmallJob), hipMemcpyHostToDevice) != hipSuccess) { rc = PRL_ERR_HIP; break; }
        prl_launch_small_iterations_many(d_jobs, n_solvers, lds_max, s0->stream);
        if (hipStreamSynchronize(s0->stream) != hipSuccess) { rc = PRL_ERR_HIP; break; }
        done += k;
    }
    (void)hipFree(d_jobs);
    if (rc != PRL_OK) { prl_set_error("HIP error in prl_solver_iterations_many"); return rc; }
    for (int i = 0; i < n_solvers; ++i) {
        prl_solver* s = solvers[i];
        s->iter += n;
        s->src[0] = s->src[1] = PRL_SRC_REGRET;
        s->ev_valid = true;
    }
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

#if !defined(PRL_EMU)
// LEVELS engine: n iterations as n replays of one captured graph. The iteration counter, the CFR+ averaging weights and the
// exploitability-history slot are read from device memory (PrlIterDev), so the captured launches never change.
static int levels_graph_iterations(prl_solver* s, int n) {
    TRY(ensure_ev(s));  // an iteration starts from the evaluation that closed the previous one
    TRY(ensure_hist(s, s->iter + n + 1));
    if (!s->d_ip) TRY(dev_alloc(s, &s->d_ip, (size_t)1));
    PrlIterDev ip;
    memset(&ip, 0, sizeof(ip));
    ip.iter = s->iter;
    ip.hist = s->d_expl_hist;
    PRL_HIP_TRY(hipMemcpyAsync(s->d_ip, &ip, sizeof(ip), hipMemcpyHostToDevice, s->stream));
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));  // `ip` is a stack variable
    if (!s->levels_graph_exec) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        PRL_HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
        prl_launch_iter_begin(s->d_ip, s->variant, s->delay, s->stream);
        for (int p = 0; p < 2; ++p) {
            if (p == 1) prl_launch_ev(s->T, s->S, s->ft.level_start.data(), s->d_term_nodes, s->n_term, s->stream);
            prl_launch_regret_strategy_dev(s->T, s->S, s->d_nodes_p[p], s->n_nodes_p[p], p, s->variant, s->d_ip, s->stream);
            prl_launch_reach(s->T, s->S, s->ft.level_start.data(), s->stream);
            prl_launch_average_dev(s->T, s->S, s->d_nodes_p[p], s->n_nodes_p[p], p, s->variant, s->d_ip, s->stream);
        }
        prl_launch_ev(s->T, s->S, s->ft.level_start.data(), s->d_term_nodes, s->n_term, s->stream);
        prl_launch_iter_end(s->d_ip, s->S.expl, s->stream);
        hipError_t e = hipStreamEndCapture(s->stream, &graph);
        if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (e != hipSuccess) {  // no graph support here: plain launches from now on
            (void)hipGetLastError();
            s->graphs_off = true;
            return PRL_ERR_UNSUPPORTED;
        }
        s->levels_graph_exec = (void*)exec;
    }
    for (int i = 0; i < n; ++i) PRL_HIP_TRY(hipGraphLaunch((hipGraphExec_t)s->levels_graph_exec, s->stream));
    s->iter += n;
    s->src[0] = s->src[1] = PRL_SRC_REGRET;
    s->ev_valid = true;
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}
#endif

int32_t prl_solver_iteration(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    return prl_solver_iterations(s, 1);
}

// n iterations. FUSED: the evaluation (both seats + best response) of the strategy after iteration t is folded into the
// first board pass of iteration t + 1, which reads the same regrets; only the last iteration runs a separate evaluation
// pass. The exploitability history is the same as n single calls produce.
int32_t prl_solver_iterations(prl_solver_t* s, int32_t n) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    if (!s->fused && s->small_tree && n > 0) return small_tree_iterations(s, n);
#if !defined(PRL_EMU)
    if (!s->fused && !s->graphs_off && n > 0) {
        const int rc = levels_graph_iterations(s, n);
        if (rc != PRL_ERR_UNSUPPORTED) return rc;
    }
#endif
    for (int i = 0; i < n; ++i) TRY(iteration_core(s, i == n - 1));
    return PRL_OK;
}

// n iterations bracketed by HIP events on the solver's own stream (what bench.py's roofline figure is derived from)
int32_t prl_solver_time_iterations_ex(prl_solver_t* s, int32_t n, float* out_ms, float* out_pass_ms, int32_t* out_n_pass) {
    if (!s || !out_ms || n < 0) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    TRY(ensure_hist(s, s->iter + n + 1));
    s->time_passes = s->fused && out_pass_ms != nullptr;
    PRL_HIP_TRY(hipEventRecord(e0, s->stream));
    int rc = prl_solver_iterations(s, n);
    s->time_passes = false;
    if (rc == PRL_OK) {
        PRL_HIP_TRY(hipEventRecord(e1, s->stream));
        PRL_HIP_TRY(hipEventSynchronize(e1));
        PRL_HIP_TRY(hipEventElapsedTime(out_ms, e0, e1));
        float pass_ms = 0.f;
        const bool dump = getenv("PRL_DUMP_PASS_MS") != nullptr;  // diagnostics: every board-pass launch's duration on stderr
        for (size_t i = 0; i + 1 < s->pass_events.size(); i += 2) {
            float t = 0.f;
            PRL_HIP_TRY(hipEventElapsedTime(&t, s->pass_events[i], s->pass_events[i + 1]));
            pass_ms += t;
            if (dump) fprintf(stderr, "%.3f%s", t, (i / 2) % 16 == 15 ? "\n" : " ");
        }
        if (dump) fprintf(stderr, "\n");
        if (out_pass_ms) *out_pass_ms = pass_ms;
        if (out_n_pass) *out_n_pass = (int32_t)(s->pass_events.size() / 2);
    }
    for (hipEvent_t e : s->pass_events) (void)hipEventDestroy(e);
    s->pass_events.clear();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// n x (update_reach + compute_ev) of the strategy the solver currently holds (LocalBRMaster.evaluate's work per call, minus the
// agent query), timed like prl_solver_time_iterations_ex: total device ms, summed board-pass kernel ms, number of launches
int32_t prl_solver_time_evaluations(prl_solver_t* s, int32_t n, float* out_ms, float* out_pass_ms, int32_t* out_n_pass) {
    if (!s || !out_ms || n < 0) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    s->time_passes = s->fused && out_pass_ms != nullptr;
    PRL_HIP_TRY(hipEventRecord(e0, s->stream));
    int rc = PRL_OK;
    for (int i = 0; i < n && rc == PRL_OK; ++i) {
        s->ev_valid = false;
        rc = do_update_reach(s, s->S);
        if (rc == PRL_OK) rc = ensure_ev(s);
    }
    s->time_passes = false;
    if (rc == PRL_OK) {
        PRL_HIP_TRY(hipEventRecord(e1, s->stream));
        PRL_HIP_TRY(hipEventSynchronize(e1));
        PRL_HIP_TRY(hipEventElapsedTime(out_ms, e0, e1));
        float pass_ms = 0.f;
        for (size_t i = 0; i + 1 < s->pass_events.size(); i += 2) {
            float t = 0.f;
            PRL_HIP_TRY(hipEventElapsedTime(&t, s->pass_events[i], s->pass_events[i + 1]));
            pass_ms += t;
        }
        if (out_pass_ms) *out_pass_ms = pass_ms;
        if (out_n_pass) *out_n_pass = (int32_t)(s->pass_events.size() / 2);
    }
    for (hipEvent_t e : s->pass_events) (void)hipEventDestroy(e);
    s->pass_events.clear();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int32_t prl_solver_time_iterations(prl_solver_t* s, int32_t n, float* out_ms) {
    return prl_solver_time_iterations_ex(s, n, out_ms, nullptr, nullptr);
}

#ifdef FHP_EXPERIMENT
// experiment builds (scripts/gpu_toggle.py): run-time switch between two code paths of the board pass
extern "C" int32_t prl_debug_set_experiment(prl_solver_t* s, int32_t flags) {
    if (!s) return PRL_ERR_ARG;
    s->fp.exp = flags;
    return PRL_OK;
}
#endif

#ifdef PRL_ST_TIMING
// instrumented builds only (scripts/st_phase_timing.py): shader clocks wave 0 of every workgroup spent per phase of the last street's pass
extern "C" int32_t prl_debug_st_timing(prl_solver_t* s, unsigned long long* out8, int32_t reset) {
    if (!s || !s->sp.timing) return PRL_ERR_ARG;
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    PRL_HIP_TRY(hipMemcpy(out8, s->sp.timing, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) PRL_HIP_TRY(hipMemset(s->sp.timing, 0, 8 * sizeof(unsigned long long)));
    return PRL_OK;
}
#endif

#ifdef PRL_FHP_TIMING
// instrumented builds only (scripts/gpu_phases.sh): shader clocks wave 0 spent per phase, summed over boards and passes
extern "C" int32_t prl_debug_fhp_timing(prl_solver_t* s, unsigned long long* out8, int32_t reset) {
    if (!s || !s->fp.timing) return PRL_ERR_ARG;
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    PRL_HIP_TRY(hipMemcpy(out8, s->fp.timing, 72 * sizeof(unsigned long long), hipMemcpyDeviceToHost));  // [8] phases + [5][12] barrier waits
    if (reset) PRL_HIP_TRY(hipMemset(s->fp.timing, 0, 72 * sizeof(unsigned long long)));
    return PRL_OK;
}
#endif

int32_t prl_solver_sync(prl_solver_t* s) {
    if (!s) { prl_set_error("NULL solver"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

// root exploitability [seat 0, seat 1] of the CURRENT strategy, raw float32 as in node.exploitability (ValueFiller.py:101)
int32_t prl_solver_exploitability(prl_solver_t* s, float* out2) {
    if (!s || !out2) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    TRY(ensure_ev(s));
    PRL_HIP_TRY(hipMemcpyAsync(out2, s->S.expl, 2 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    return prl_solver_sync(s);
}

// _CFRBase._evaluate_avg_strats (_CFRBase.py:218-262): exploitability of the average strategy; training state untouched
int32_t prl_solver_eval_avg(prl_solver_t* s, float* out2) {
    if (!s || !out2) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (!s->eval_ready) {
        TRY(alloc_node_vectors(s, &s->Seval, false));
        s->eval_ready = true;
    }
    TRY(ensure_board_avg(s));
    PrlDevState E = s->Seval;
    E.strategy = s->d_avg;  // trunk columns precede the board columns, so the trunk view indexes the same array
    E.strat_f64 = s->S.avg_f64;
    E.regret = nullptr; E.avg = nullptr; E.avg_sum = nullptr; E.avg_f64 = nullptr;
    TRY(do_update_reach(s, E));
    if (s->fused) {
        if (s->avg_f32) {
            // float32 storage: a blended average is widened and played with float64 arithmetic (what ARR64 does with the float64 one); before
            // the first blend the average IS a float32 strategy: played as is, float32 arithmetic (what ARR32 does)
            const int src = s->board_avg_f64 ? PRL_SRC_AVGF32 : PRL_SRC_STRAT32;
            TRY(fused_board_pass(s, E, PRL_FHP_EVAL, src, src, nullptr, s->board_avg_f64 ? nullptr : s->d_avg32));
        } else {
            const int src = s->board_avg_f64 ? PRL_SRC_ARR64 : PRL_SRC_ARR32;
            TRY(fused_board_pass(s, E, PRL_FHP_EVAL, src, src, s->d_avg));
        }
    }
    prl_launch_ev(s->T, E, s->ft.level_start.data(), s->d_term_nodes, s->n_term, s->stream);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipMemcpyAsync(out2, E.expl, 2 * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    return prl_solver_sync(s);
}

int32_t prl_solver_get_cols(prl_solver_t* s, int32_t field, int64_t col_begin, int64_t n_cols, void* out) {
    if (!s || !out || col_begin < 0 || n_cols < 0 || col_begin + n_cols > s->full_cols) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    const char* src = nullptr;
    size_t elem = 0;
    double fill[PRL_FHP_MAX_NODES * 3] = {0.};
    if (!s->col_dfs.empty()) {
        // the per-street engine keeps its columns in an internal order (trunk, then street by street, instance by instance): every requested flat-tree
        // column is fetched from where it lives; runs of columns that are neighbours in both orders (a node's actions) travel as one copy
        switch (field) {
            case PRL_SF_REGRET: src = (const char*)s->d_regret; elem = 4; break;
            case PRL_SF_AVG:
                if (s->avg_f32) { prl_set_error("get_cols(AVG): the average is stored as float32 in this solver; use prl_solver_get"); return PRL_ERR_UNSUPPORTED; }
                TRY(ensure_board_avg(s)); src = (const char*)s->d_avg; elem = 8; break;
            case PRL_SF_AVG_SUM: src = (const char*)s->S.avg_sum; elem = 4; break;
            default: prl_set_error("get_cols: REGRET, AVG or AVG_SUM"); return PRL_ERR_ARG;
        }
        if (!src) { prl_set_error("field not available for this variant"); return PRL_ERR_STATE; }
        if (s->col_int.empty()) {
            s->col_int.resize(s->col_dfs.size());
            for (size_t c = 0; c < s->col_dfs.size(); ++c) s->col_int[s->col_dfs[c]] = (int32_t)c;
        }
        const size_t cb = (size_t)s->R * elem;
        for (int64_t c = col_begin; c < col_begin + n_cols;) {
            int64_t run = 1;
            while (c + run < col_begin + n_cols && s->col_int[c + run] == s->col_int[c] + (int32_t)run) ++run;
            PRL_HIP_TRY(hipMemcpyAsync((char*)out + (size_t)(c - col_begin) * cb, src + (size_t)s->col_int[c] * cb, (size_t)run * cb, hipMemcpyDeviceToHost, s->stream));
            c += run;
        }
        return prl_solver_sync(s);
    }
    switch (field) {
        case PRL_SF_REGRET: src = (const char*)s->d_regret; elem = 4; break;
        case PRL_SF_AVG:
            if (s->avg_f32) { prl_set_error("get_cols(AVG): the average is stored as float32 in this solver; use prl_solver_get"); return PRL_ERR_UNSUPPORTED; }
            TRY(ensure_board_avg(s)); src = (const char*)s->d_avg; elem = 8;
            if (s->sorted) blocked_avg_fill(s, fill);
            break;
        case PRL_SF_AVG_SUM: src = (const char*)s->S.avg_sum; elem = 4; break;
        default: prl_set_error("get_cols: REGRET, AVG or AVG_SUM"); return PRL_ERR_ARG;
    }
    if (!src) { prl_set_error("field not available for this variant"); return PRL_ERR_STATE;}
    const size_t cb = (size_t)s->R * elem;
    if (!s->sorted) {
        PRL_HIP_TRY(hipMemcpyAsync(out, src + (size_t)col_begin * cb, (size_t)n_cols * cb, hipMemcpyDeviceToHost, s->stream));
        return prl_solver_sync(s);
    }
    // sorted storage: the trunk's columns as they are, the boards' columns translated board by board (whole boards through the staging buffer)
    int64_t c = col_begin;
    const int64_t c_end = col_begin + n_cols;
    char* o = (char*)out;
    if (c < s->col_base) {
        const int64_t n = (c_end < s->col_base ? c_end : (int64_t)s->col_base) - c;
        PRL_HIP_TRY(hipMemcpyAsync(o, src + (size_t)c * cb, (size_t)n * cb, hipMemcpyDeviceToHost, s->stream));
        o += (size_t)n * cb; c += n;
    }
    if (c < c_end) {
        const int b_first = (int)((c - s->col_base) / s->ncb), b_last = (int)((c_end - 1 - s->col_base) / s->ncb);
        const bool whole = (c - s->col_base) % s->ncb == 0 && (c_end - s->col_base) % s->ncb == 0;
        const char* region = src + s->board_ofs * elem;
        if (whole) TRY(sorted_get_boards(s, region, (int)elem, fill, nullptr, o, (int)elem, b_first, b_last - b_first + 1));
        else {
            std::vector<char> tmp((size_t)(b_last - b_first + 1) * s->ncb * cb);
            TRY(sorted_get_boards(s, region, (int)elem, fill, nullptr, tmp.data(), (int)elem, b_first, b_last - b_first + 1));
            memcpy(o, tmp.data() + (size_t)(c - s->col_base - (int64_t)b_first * s->ncb) * cb, (size_t)(c_end - c) * cb);
        }
    }
    return prl_solver_sync(s);
}

int32_t prl_solver_get(prl_solver_t* s, int32_t field, void* out) {
    if (!s || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    const size_t nv = (size_t)s->T.n_nodes * 2 * s->T.R, nc = (size_t)s->full_cols * s->R;
    const void* src = nullptr;
    size_t bytes = 0;
    const bool node_vectors = field == PRL_SF_REACH || field == PRL_SF_EV || field == PRL_SF_EV_BR || field == PRL_SF_BR_IDX ||
                              field == PRL_SF_STRAT_F64 || field == PRL_SF_AVG_F64;
    if (s->fused && node_vectors) {
        prl_set_error("the fused engine keeps per-node vectors on chip; create the solver with PRL_ENGINE_LEVELS to read them");
        return PRL_ERR_UNSUPPORTED;
    }
    switch (field) {
        case PRL_SF_REACH: src = s->S.reach; bytes = nv * 4; break;
        case PRL_SF_EV: TRY(ensure_ev(s)); src = s->S.ev; bytes = nv * 4; break;
        case PRL_SF_EV_BR: TRY(ensure_ev(s)); src = s->S.ev_br; bytes = nv * 4; break;
        case PRL_SF_STRATEGY:
            if (s->sorted) {
                const size_t tcb = (size_t)s->T.n_cols * s->R;
                double* o = (double*)out;
                double fill[PRL_FHP_MAX_NODES * 3];
                if (s->user_strategy_f64 == 1) {
                    PRL_HIP_TRY(hipMemcpyAsync(o, s->d_user_strategy, tcb * 8, hipMemcpyDeviceToHost, s->stream));
                    TRY(sorted_get_boards(s, s->d_user_strategy + s->board_ofs, 8, nullptr, s->d_user_blocked64, o + tcb, 8, 0, s->fp.n_boards));
                    return prl_solver_sync(s);
                }
                if (s->user_strategy_f64 == 0) {  // stored as float32: widened on the way out (exact)
                    std::vector<float> f(tcb);
                    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
                    PRL_HIP_TRY(hipMemcpy(f.data(), s->d_user_strategy32, tcb * 4, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < tcb; ++i) o[i] = (double)f[i];
                    TRY(sorted_get_boards(s, s->d_user_strategy32 + s->board_ofs, 4, nullptr, s->d_user_blocked32, o + tcb, 8, 0, s->fp.n_boards));
                    return prl_solver_sync(s);
                }
                if (s->src[0] != PRL_SRC_REGRET || s->src[1] != PRL_SRC_REGRET) {  // uniform float64 fill
                    prl_set_error("fused engine: strategy is the implicit uniform fill until both seats have been updated");
                    return PRL_ERR_STATE;
                }
                if (!s->d_user_strategy) TRY(dev_alloc(s, &s->d_user_strategy, col_array_elems(s)));  // (scratch: user_strategy_f64 stays -1)
                PrlFhpParams fp = s->fp;
                fp.variant = s->variant;
                fp.regret = s->d_regret + s->board_ofs;
                prl_launch_fhp_strategy_from_regret(fp, s->d_user_strategy + s->board_ofs, s->stream);
                PRL_HIP_TRY(hipMemcpyAsync(o, s->S.strategy, tcb * 8, hipMemcpyDeviceToHost, s->stream));
                for (int j = 0; j < s->ncb; ++j) fill[j] = (double)(float)(1.0 / (double)board_col_actions(s, j));  // all-zero regrets: uniform
                TRY(sorted_get_boards(s, s->d_user_strategy + s->board_ofs, 8, fill, nullptr, o + tcb, 8, 0, s->fp.n_boards));
                return prl_solver_sync(s);
            }
            if (s->fused) {
                if (s->user_strategy_f64 == 1) { src = s->d_user_strategy; bytes = nc * 8; break; }
                if (s->user_strategy_f64 == 0) {  // stored as float32: widened on the way out (exact)
                    std::vector<float> f(nc);
                    PRL_HIP_TRY(hipStreamSynchronize(s->stream));
                    PRL_HIP_TRY(hipMemcpy(f.data(), s->d_user_strategy32, nc * 4, hipMemcpyDeviceToHost));
                    double* o = (double*)out;
                    if (s->col_dfs.empty()) for (size_t i = 0; i < nc; ++i) o[i] = (double)f[i];
                    else
                        for (int c = 0; c < s->full_cols; ++c)
                            for (int h = 0; h < s->R; ++h) o[(size_t)s->col_dfs[c] * s->R + h] = (double)f[(size_t)c * s->R + h];
                    return PRL_OK;
                }
                if (s->src[0] != PRL_SRC_REGRET || s->src[1] != PRL_SRC_REGRET) {  // uniform float64 fill
                    prl_set_error("fused engine: strategy is the implicit uniform fill until both seats have been updated");
                    return PRL_ERR_STATE;
                }
                if (!s->d_user_strategy) TRY(dev_alloc(s, &s->d_user_strategy, nc));
                PRL_HIP_TRY(hipMemcpyAsync(s->d_user_strategy, s->S.strategy, (size_t)s->T.n_cols * s->R * 8, hipMemcpyDeviceToDevice, s->stream));
                if (s->streets) {
                    for (int lv = 0; lv < s->st.n_groups; ++lv) {
                        PrlStParams q = s->sp;
                        q.n_inst = s->st.group[lv].n_inst; q.col_base = s->st.group[lv].col_base; q.regret = s->d_regret; q.variant = s->variant;
                        prl_launch_st_strategy_from_regret(q, s->st.group[lv].spec, s->d_user_strategy, s->stream);
                    }
                } else {
                    PrlFhpParams fp = s->fp;
                    fp.variant = s->variant;
                    prl_launch_fhp_strategy_from_regret(fp, s->d_user_strategy, s->stream);
                }
                src = s->d_user_strategy; bytes = nc * 8;
                break;
            }
            src = s->S.strategy; bytes = nc * 8; break;
        case PRL_SF_STRAT_F64: src = s->S.strat_f64; bytes = (size_t)s->T.n_nodes; break;
        case PRL_SF_REGRET:
            if (s->sorted) {
                const size_t tcb = (size_t)s->T.n_cols * s->R;
                PRL_HIP_TRY(hipMemcpyAsync(out, s->d_regret, tcb * 4, hipMemcpyDeviceToHost, s->stream));
                TRY(sorted_get_boards(s, s->d_regret + s->board_ofs, 4, nullptr, nullptr, (float*)out + tcb, 4, 0, s->fp.n_boards));
                return prl_solver_sync(s);
            }
            src = s->d_regret; bytes = nc * 4; break;
        case PRL_SF_AVG:
            if (s->sorted) {  // trunk columns float64; board columns float64, or float32 widened (exact)
                const size_t tcb = (size_t)s->T.n_cols * s->R;
                double fill[PRL_FHP_MAX_NODES * 3];
                TRY(ensure_board_avg(s));
                blocked_avg_fill(s, fill);
                PRL_HIP_TRY(hipMemcpyAsync(out, s->d_avg, tcb * 8, hipMemcpyDeviceToHost, s->stream));
                if (s->avg_f32) TRY(sorted_get_boards(s, s->d_avg32 + s->board_ofs, 4, fill, nullptr, (double*)out + tcb, 8, 0, s->fp.n_boards));
                else TRY(sorted_get_boards(s, s->d_avg + s->board_ofs, 8, fill, nullptr, (double*)out + tcb, 8, 0, s->fp.n_boards));
                return prl_solver_sync(s);
            }
            if (s->avg_f32 && s->streets) {  // trunk columns float64, street columns float32 widened (exact); internal column order -> the flat tree's
                const size_t tcb = (size_t)s->T.n_cols * s->R;
                std::vector<float> f(nc);
                std::vector<double> tr(tcb);
                PRL_HIP_TRY(hipStreamSynchronize(s->stream));
                PRL_HIP_TRY(hipMemcpy(f.data(), s->d_avg32, nc * 4, hipMemcpyDeviceToHost));
                PRL_HIP_TRY(hipMemcpy(tr.data(), s->d_avg, tcb * 8, hipMemcpyDeviceToHost));
                double* o = (double*)out;
                for (int c = 0; c < s->full_cols; ++c) {
                    double* dst = o + (size_t)(s->col_dfs.empty() ? c : s->col_dfs[c]) * s->R;
                    if (c < s->T.n_cols) memcpy(dst, tr.data() + (size_t)c * s->R, (size_t)s->R * 8);
                    else for (int h = 0; h < s->R; ++h) dst[h] = (double)f[(size_t)c * s->R + h];
                }
                return PRL_OK;
            }
            TRY(ensure_board_avg(s)); src = s->d_avg; bytes = nc * 8; break;
        case PRL_SF_AVG_F64: src = s->S.avg_f64; bytes = (size_t)s->T.n_nodes; break;
        case PRL_SF_AVG_SUM:
            if (s->sorted && s->S.avg_sum) {
                const size_t tcb = (size_t)s->T.n_cols * s->R;
                PRL_HIP_TRY(hipMemcpyAsync(out, s->S.avg_sum, tcb * 4, hipMemcpyDeviceToHost, s->stream));
                TRY(sorted_get_boards(s, s->S.avg_sum + s->board_ofs, 4, nullptr, nullptr, (float*)out + tcb, 4, 0, s->fp.n_boards));
                return prl_solver_sync(s);
            }
            src = s->S.avg_sum; bytes = nc * 4; break;
        case PRL_SF_BR_IDX: TRY(ensure_ev(s)); src = s->S.br_idx; bytes = (size_t)s->T.n_nodes * s->T.R * 4; break;
        case PRL_SF_EXPL_HISTORY: src = s->d_expl_hist; bytes = (size_t)(s->iter + 1) * 2 * 4; break;
        case PRL_SF_ITER: *(int32_t*)out = s->iter; return PRL_OK;
        case PRL_SF_CONSTANTS: ((float*)out)[0] = s->T.chance_prob; ((float*)out)[1] = s->T.eq_const; return PRL_OK;
        case PRL_SF_BYTES_ALLOCATED: *(int64_t*)out = (int64_t)s->bytes_allocated; return PRL_OK;
        case PRL_SF_ENGINE: *(int32_t*)out = s->fused ? PRL_ENGINE_FUSED : PRL_ENGINE_LEVELS; return PRL_OK;
        case PRL_SF_GRAPH_REPLAY: *(int32_t*)out = s->levels_graph_exec != nullptr; return PRL_OK;
        case PRL_SF_EXCHANGES: *(int64_t*)out = (int64_t)s->n_exchanges; return PRL_OK;
        case PRL_SF_VMM_RANGES: {
            int64_t bytes = 0;
#if !defined(PRL_EMU)
            for (void* r : s->vmm) bytes += (int64_t)((PrlVmmRange*)r)->size;
#endif
            ((int64_t*)out)[0] = (int64_t)s->vmm.size(); ((int64_t*)out)[1] = bytes; return PRL_OK;
        }
        case PRL_SF_EXPLICIT_STRATEGY: *(int32_t*)out = s->fused ? s->user_strategy_f64 : -1; return PRL_OK;
        default: prl_set_error("unknown solver field"); return PRL_ERR_ARG;
    }
    if (!src) { prl_set_error("field not available for this variant"); return PRL_ERR_STATE; }
    const bool col_field = field == PRL_SF_STRATEGY || field == PRL_SF_REGRET || field == PRL_SF_AVG || field == PRL_SF_AVG_SUM;
    if (col_field && !s->col_dfs.empty()) {  // internal column order -> the flat tree's DFS column order
        std::vector<char> tmp(bytes);
        PRL_HIP_TRY(hipMemcpyAsync(tmp.data(), src, bytes, hipMemcpyDeviceToHost, s->stream));
        TRY(prl_solver_sync(s));
        const size_t cb = bytes / (size_t)s->full_cols;
        for (int c = 0; c < s->full_cols; ++c) memcpy((char*)out + (size_t)s->col_dfs[c] * cb, tmp.data() + (size_t)c * cb, cb);
        return PRL_OK;
    }
    PRL_HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, s->stream));
    return prl_solver_sync(s);
}

}  // extern "C"

// ---- the policy table of a solver's average strategy (prl_policy.h; include/pokerrl_hip.h section 6) --------------------------------------------------
namespace {
// columns [.][R] of `elem` bytes (4: float32, 8: float64) -> rows of the table. Row r of this launch is row row0 + r of the table; rows repeat with
// `period` (the decision nodes of one board subtree; period = the number of rows when nothing repeats): row r is entry t = r % period of the per-row
// arrays in repetition rep = r / period, its j-th action column is source column rep * cols_per_rep + col0[t] + j and goes to action col_action[col0[t] + j].
// row_id (may be null): the table row of entry r is row0 + row_id[r] instead of row0 + r (a subset of the rows: the per-street engine's trunk / street rows).
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(256) prl_k_table_fill(const void* cols, int elem, int R, int n_rows_here, int period, int cols_per_rep, const int32_t* col0,
                                                        const int32_t* nch, const int32_t* col_action, int n_actions, long long row0, float* probs, const int32_t* row_id) {
    const long long i = (long long)prl_bid() * prl_nthreads() + prl_tid();
    if (i >= (long long)n_rows_here * R) return;
    const int r = (int)(i / R), h = (int)(i - (long long)r * R);
    const int rep = r / period, t = r - rep * period;
    const long long row = row0 + (row_id ? row_id[r] : r);
    for (int j = 0; j < nch[t]; ++j) {
        const size_t at = ((size_t)rep * cols_per_rep + col0[t] + j) * R + h;
        const float v = elem == 8 ? (float)((const double*)cols)[at] : ((const float*)cols)[at];
        probs[((size_t)row * n_actions + col_action[col0[t] + j]) * R + h] = v;
    }
}

struct TableReplay {
    const PrlFlatTree& t;
    uint32_t seed;
    std::vector<LbrbHistKey> key_of;  // per node (decision nodes)
    std::vector<int32_t> twin_of;     // node whose public state and history this node repeats (two bet sizes that the env turns into one amount), else -1
    std::vector<int32_t> rows;        // decision nodes that get a row, in node order
    TableReplay(const PrlFlatTree& tree, uint32_t key_seed) : t(tree), seed(key_seed), key_of(tree.n_nodes), twin_of(tree.n_nodes, -1) {}
    void walk(int id, const PrlEnvState& st, const LbrbHistKey& parent_key) {
        const int8_t* row = t.board_id[id] >= 0 ? t.boards.data() + (size_t)t.board_id[id] * t.board_len : nullptr;
        int n_dealt = 0;
        for (int c = 0; row && c < t.board_len; ++c) n_dealt += row[c] >= 0;
        int8_t none[PRL_MAX_BOARD_CARDS] = {0, 0, 0, 0, 0};
        key_of[id] = lbrb_hist_step(parent_key, st, row ? row : none, n_dealt, t.rules.n_board_cards, t.rules.n_suits);
        if (t.n_children[id] == 0) return;
        rows.push_back(id);
        for (int i = 0; i < t.n_children[id]; ++i) {
            const int c = t.child_list[t.child_start[id] + i];
            PrlEnvState s2 = st;
            PrlStepInfo info;
            prl_env_step(t.game, s2, t.col_action[t.first_col[id] + i], &info);
            if (info.is_terminal) continue;  // fold / showdown leaves, all-in run-out chains: no decisions below
            if (t.kind[c] == PRL_NODE_CHANCE) {
                for (int k = 0; k < t.n_children[c]; ++k) walk(t.child_list[t.child_start[c] + k], s2, key_of[id]);
            } else walk(c, s2, key_of[id]);
        }
    }
};
}  // namespace

// ---- an agent's strategy from a DEVICE array (StrategyFiller.py:88-116 without the host in between) -----------------------------------------------------
namespace {
// probs [n_dec][R][A] float32 (decision nodes in node order, every action int) -> action columns. Column c of this launch is entry c of the maps:
// node ordinal col_k[c], action col_a[c]; dst32 / dst64 (either may be null) take it at dst_col0 + c.
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(256) prl_k_cols_from_node_probs(const float* probs, int R, int A, const int32_t* col_k, const int32_t* col_a, long long n_cols_here,
                                                                  float* dst32, double* dst64) {
    const long long i = (long long)prl_bid() * prl_nthreads() + prl_tid();
    if (i >= n_cols_here * R) return;
    const long long c = i / R;
    const int h = (int)(i - c * R);
    const float v = probs[((size_t)col_k[c] * R + h) * A + col_a[c]];
    if (dst32) dst32[(size_t)c * R + h] = v;
    if (dst64) dst64[(size_t)c * R + h] = (double)v;
}
}  // namespace

extern "C" int32_t prl_solver_set_strategy_device(prl_solver_t* s, const prl_tree_t* tree, const float* d_probs, int32_t n_actions) {
    if (!s || !tree || !d_probs || n_actions < 2) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    const PrlFlatTree& full = *prl_tree_flat(tree);
    if (full.n_nodes != s->full_nodes || full.n_cols != s->full_cols || full.rules.range_size != s->R) { prl_set_error("prl_solver_set_strategy_device: not the tree this solver was created on"); return PRL_ERR_ARG; }
    // per column (in the engine's internal column order): the ordinal of its decision node among the decision nodes, its action int
    std::vector<int32_t> ord(full.n_nodes, -1), maps((size_t)2 * s->full_cols);
    int n_dec = 0;
    for (int n = 0; n < full.n_nodes; ++n)
        if (full.kind[n] == PRL_NODE_DECISION) ord[n] = n_dec++;
    for (int ci = 0; ci < s->full_cols; ++ci) {
        const int c = s->col_dfs.empty() ? ci : s->col_dfs[ci];
        if (full.col_action[c] < 0 || full.col_action[c] >= n_actions) { prl_set_error("prl_solver_set_strategy_device: the tree has an action outside [0, n_actions)"); return PRL_ERR_ARG; }
        maps[ci] = ord[full.col_node[c]];
        maps[(size_t)s->full_cols + ci] = full.col_action[c];
    }
    int32_t* d_maps = nullptr;
    PRL_HIP_TRY(hipMalloc((void**)&d_maps, maps.size() * sizeof(int32_t)));
    auto fail = [&](int code) { (void)hipFree(d_maps); return code; };
#define SD_TRY(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); prl_set_error("HIP error in prl_solver_set_strategy_device"); return fail(PRL_ERR_HIP); } } while (0)
    SD_TRY(hipMemcpy(d_maps, maps.data(), maps.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    const int32_t *d_k = d_maps, *d_a = d_maps + s->full_cols;
    const int R = s->R;
    auto launch = [&](long long c0, long long n, float* d32, double* d64) {
        if (n <= 0) return;
        PRL_LAUNCH(prl_k_cols_from_node_probs, (unsigned)((n * R + 255) / 256), 256, 0, s->stream, d_probs, R, (int)n_actions, d_k + c0, d_a + c0, n, d32, d64);
    };
    const long long tc = s->T.n_cols;  // LEVELS: every column; FUSED: the trunk's (they precede the others)
    launch(0, tc, nullptr, s->S.strategy);
    if (s->sorted) {
        const size_t ne = col_array_elems(s), nblk = (size_t)s->fp.n_boards * s->ncb * PRL_FHP_NBLOCKED;
        int rc = PRL_OK;
        if (!s->d_user_strategy32) rc = dev_alloc(s, &s->d_user_strategy32, ne);
        if (!rc && !s->d_user_blocked32) rc = dev_alloc(s, &s->d_user_blocked32, nblk);
        if (!rc) rc = stage_ready(s);
        if (rc) return fail(rc);
        launch(0, tc, s->d_user_strategy32, nullptr);
        for (int at = 0; at < s->fp.n_boards; at += s->stage_boards) {  // the boards' columns: hand order in the staging buffer, then into sorted storage
            const int nb = s->fp.n_boards - at < s->stage_boards ? s->fp.n_boards - at : s->stage_boards;
            launch((long long)s->col_base + (long long)at * s->ncb, (long long)nb * s->ncb, (float*)s->d_stage, nullptr);
            prl_launch_fhp_compact(s->fp, s->d_stage, 4, at, nb, s->d_user_strategy32 + s->board_ofs, s->d_user_blocked32, s->stream);
            SD_TRY(hipGetLastError());
        }
        s->user_strategy_f64 = 0;
    } else if (s->fused) {
        if (!s->d_user_strategy32) { const int rc = dev_alloc(s, &s->d_user_strategy32, (size_t)s->full_cols * R); if (rc) return fail(rc); }
        launch(0, s->full_cols, s->d_user_strategy32, nullptr);
        s->user_strategy_f64 = 0;
    }
    SD_TRY(hipGetLastError());
    SD_TRY(hipMemsetAsync(s->S.strat_f64, 0, (size_t)s->T.n_nodes, s->stream));  // a float32 strategy: float32 arithmetic at every node
    SD_TRY(hipStreamSynchronize(s->stream));
#undef SD_TRY
    (void)hipFree(d_maps);
    s->ev_valid = false;
    return do_update_reach(s, s->S);
}

extern "C" int32_t prl_policy_table_from_solver(prl_solver_t* s, const prl_tree_t* tree, uint32_t key_seed, prl_policy_table_t** out) {
    if (!s || !tree || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    *out = nullptr;
    const PrlFlatTree& full = *prl_tree_flat(tree);
    if (full.n_nodes != s->full_nodes || full.n_cols != s->full_cols || full.rules.range_size != s->R) { prl_set_error("prl_policy_table_from_solver: not the tree this solver was created on"); return PRL_ERR_ARG; }
    if (s->world > 1) { prl_set_error("prl_policy_table_from_solver: an unsharded solve (a rank of a sharded one holds its share of the boards only)"); return PRL_ERR_UNSUPPORTED; }
    if (s->fused && s->user_strategy_f64 >= 0) { prl_set_error("prl_policy_table_from_solver: an explicit strategy is loaded; the table is made of a CFR run's average"); return PRL_ERR_STATE; }
    if (s->iter < 1 || (s->variant == PRL_CFR_PLUS && s->iter <= s->delay)) { prl_set_error("prl_policy_table_from_solver: no average strategy yet (iterate first; CFR+ starts averaging after `delay` iterations)"); return PRL_ERR_STATE; }
    if (s->fused) {  // a run of prl_solver_iterations leaves its last evaluation (and the Vanilla / Linear average update riding on it) pending
        TRY(ensure_ev(s));
        if (s->avg_pending[0] >= 0 || s->avg_pending[1] >= 0) { prl_set_error("prl_policy_table_from_solver: iteration not closed"); return PRL_ERR_STATE; }
    }
    TRY(ensure_board_avg(s));
    // rows and their history keys: the env replayed along the tree
    TableReplay rp(full, key_seed);
    {
        PrlEnvState st;
        prl_env_reset(full.game, st);
        rp.walk(0, st, lbrb_hist_root(key_seed));
    }
    const int n_act = full.game.game_type == PRL_GAME_LIMIT ? 3 : full.game.n_bet_sizes + 2;
    // two rows with one key: children of one node that the env turns into the same state (a bet size below the minimum raise and the minimum raise
    // itself) repeat each other node for node -- the first one keeps the rows (no look-up could tell them apart); anything else is a hash collision
    std::vector<int32_t> rows;
    {
        std::unordered_map<unsigned long long, int32_t> first;  // key -> the node that holds it (rp.rows is in node order: parents come first)
        first.reserve(rp.rows.size() * 2);
        auto dec_parent = [&](int n) { int p = full.parent[n]; while (p >= 0 && full.kind[p] != PRL_NODE_DECISION) p = full.parent[p]; return p; };
        auto orig = [&](int n) { return n >= 0 && rp.twin_of[n] >= 0 ? rp.twin_of[n] : n; };
        for (int32_t b : rp.rows) {
            auto it = first.emplace(lbrb_key64(rp.key_of[b]), b);
            if (it.second) { rows.push_back(b); continue; }
            const int a = it.first->second, pa = dec_parent(a), pb = dec_parent(b);
            if (!(pa >= 0 && pb >= 0 && orig(pa) == orig(pb) && full.board_id[a] == full.board_id[b])) {
                prl_set_error("prl_policy_table_from_solver: nodes " + std::to_string(a) + " and " + std::to_string(b) + " share a 64-bit history key; build the table under another key_seed");
                return PRL_ERR_STATE;
            }
            rp.twin_of[b] = orig(a);
        }
    }
    const int n_rows = (int)rows.size();
    if (n_rows < 1) { prl_set_error("prl_policy_table_from_solver: the tree has no decision node"); return PRL_ERR_ARG; }
    uint32_t cap = 2;
    while (cap < 2u * (uint32_t)n_rows) cap *= 2;
    std::vector<uint64_t> keys(cap, 0);
    std::vector<int32_t> slot_row(cap, -1);
    for (int r = 0; r < n_rows; ++r) {
        const LbrbHistKey& hk = rp.key_of[rows[r]];
        uint32_t i = lbrb_first_slot(hk, cap - 1);
        while (keys[i] != 0) i = (i + 1u) & (cap - 1);
        keys[i] = lbrb_key64(hk); slot_row[i] = r;
    }
    PrlPolicyTable* T = prl_policy_table_alloc(keys.data(), slot_row.data(), cap, nullptr, n_rows, n_act, s->R, key_seed);
    if (!T) return PRL_ERR_OOM;
    T->suit_canon = s->symmetrize ? 1 : 0;
    int rc = PRL_OK;
    int32_t* d_meta = nullptr;
    auto fail = [&](int code) { prl_policy_table_destroy(T); (void)hipFree(d_meta); return code; };
#define PT_TRY(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); prl_set_error("HIP error in prl_policy_table_from_solver"); return fail(PRL_ERR_HIP); } } while (0)
    if (s->streets) {
        // the per-street engine: hand-order columns in the engine's INTERNAL column order (trunk, then group by group, instance by instance; a node's columns
        // stay adjacent); the street columns' average may live in the float32 array (PRL_SOLVER_AVG_F32, CFR+)
        std::vector<int32_t> inv(s->full_cols, -1), act(s->full_cols, 0);
        for (int ci = 0; ci < s->full_cols; ++ci) {
            const int c = s->col_dfs.empty() ? ci : s->col_dfs[ci];
            inv[c] = ci;
            act[ci] = full.col_action[c];
        }
        const bool street_f32 = s->avg_f32;  // (as prl_solver_get(AVG) reads them)
        const int n_trunk_cols = s->T.n_cols;
        for (int part = 0; part < 2; ++part) {  // 0: the trunk's rows (float64), 1: the streets' rows
            std::vector<int32_t> col0, nch, rid;
            for (int r = 0; r < n_rows; ++r) {
                const int c0 = inv[full.first_col[rows[r]]];
                if ((c0 >= n_trunk_cols) != (part == 1)) continue;
                col0.push_back(c0); nch.push_back(full.n_children[rows[r]]); rid.push_back(r);
            }
            const int n = (int)col0.size();
            if (n == 0) continue;
            std::vector<int32_t> meta;
            meta.insert(meta.end(), col0.begin(), col0.end());
            meta.insert(meta.end(), nch.begin(), nch.end());
            meta.insert(meta.end(), rid.begin(), rid.end());
            meta.insert(meta.end(), act.begin(), act.end());
            PT_TRY(hipMalloc((void**)&d_meta, meta.size() * 4 + 16));
            PT_TRY(hipMemcpy(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
            PT_TRY(hipStreamSynchronize(s->stream));
            const bool f32 = part == 1 && street_f32;
            const long long items = (long long)n * s->R;
            PRL_LAUNCH(prl_k_table_fill, (unsigned)((items + 255) / 256), 256, 0, s->stream, f32 ? (const void*)s->d_avg32 : (const void*)s->d_avg, f32 ? 4 : 8, s->R, n, n, 0,
                       (const int32_t*)d_meta, (const int32_t*)(d_meta + n), (const int32_t*)(d_meta + 3 * (size_t)n), n_act, 0ll, T->probs, (const int32_t*)(d_meta + 2 * (size_t)n));
            PT_TRY(hipGetLastError());
            PT_TRY(hipStreamSynchronize(s->stream));
            (void)hipFree(d_meta);
            d_meta = nullptr;
        }
        *out = T;
        return PRL_OK;
    }
    // rows whose columns lie in hand order ([col][R] float64): every row of the LEVELS engine / an unsorted fused solve, the trunk's rows of the sorted one
    int n_plain = n_rows;
    if (s->sorted) {
        n_plain = 0;
        while (n_plain < n_rows && full.first_col[rows[n_plain]] < s->col_base) ++n_plain;
        // the boards' rows: n_dec per board, board after board, columns col_base + b * ncb + the shape's local columns
        const int n_dec = s->fp.n_dec;
        bool ok = n_rows - n_plain == s->fp.n_boards * n_dec;
        for (int b = 0; ok && b < s->fp.n_boards; b += (s->fp.n_boards > 64 ? s->fp.n_boards / 64 : 1))
            for (int d = 0; d < n_dec; ++d) {
                const int n = rows[n_plain + b * n_dec + d];
                ok = ok && full.first_col[n] == s->col_base + b * s->ncb + s->fp.dec_col0[d] && full.n_children[n] == s->fp.dec_nch[d];
            }
        if (!ok) { prl_set_error("prl_policy_table_from_solver: the tree's rows do not follow the board pass's layout"); return fail(PRL_ERR_UNSUPPORTED); }
    }
    {
        std::vector<int32_t> meta;  // col0[n_plain], nch[n_plain], col_action[full trunk / all columns]
        const int n_cols_plain = s->sorted ? s->col_base : s->full_cols;
        for (int r = 0; r < n_plain; ++r) meta.push_back(full.first_col[rows[r]]);
        for (int r = 0; r < n_plain; ++r) meta.push_back(full.n_children[rows[r]]);
        for (int c = 0; c < n_cols_plain; ++c) meta.push_back(full.col_action[c]);
        if (s->sorted) {  // one board's decision nodes: local col0, nch, the local columns' actions
            for (int d = 0; d < s->fp.n_dec; ++d) meta.push_back(s->fp.dec_col0[d]);
            for (int d = 0; d < s->fp.n_dec; ++d) meta.push_back(s->fp.dec_nch[d]);
            for (int j = 0; j < s->ncb; ++j) meta.push_back(full.col_action[s->col_base + j]);
        }
        PT_TRY(hipMalloc((void**)&d_meta, meta.size() * 4 + 16));
        PT_TRY(hipMemcpy(d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
        PT_TRY(hipStreamSynchronize(s->stream));
        if (n_plain > 0) {
            const long long n = (long long)n_plain * s->R;
            PRL_LAUNCH(prl_k_table_fill, (unsigned)((n + 255) / 256), 256, 0, s->stream, (const void*)s->d_avg, 8, s->R, n_plain, n_plain, 0, (const int32_t*)d_meta,
                       (const int32_t*)(d_meta + n_plain), (const int32_t*)(d_meta + 2 * n_plain), n_act, 0ll, T->probs, (const int32_t*)nullptr);
            PT_TRY(hipGetLastError());
        }
        if (s->sorted) {
            rc = stage_ready(s);
            if (rc) return fail(rc);
            double fill[PRL_FHP_MAX_NODES * 3] = {0.};
            blocked_avg_fill(s, fill);
            const int elem = s->avg_f32 ? 4 : 8;
            const void* region = s->avg_f32 ? (const void*)(s->d_avg32 + s->board_ofs) : (const void*)(s->d_avg + s->board_ofs);
            {
                char tmp[PRL_FHP_MAX_NODES * 3 * 8];
                for (int j = 0; j < s->ncb; ++j) {
                    if (elem == 4) ((float*)tmp)[j] = (float)fill[j];
                    else ((double*)tmp)[j] = fill[j];
                }
                PT_TRY(hipMemcpyAsync(s->d_fill, tmp, (size_t)s->ncb * elem, hipMemcpyHostToDevice, s->stream));
                PT_TRY(hipStreamSynchronize(s->stream));
            }
            const int32_t* m = d_meta + 2 * n_plain + s->col_base;
            const int n_dec = s->fp.n_dec;
            for (int at = 0; at < s->fp.n_boards; at += s->stage_boards) {
                const int nb = s->fp.n_boards - at < s->stage_boards ? s->fp.n_boards - at : s->stage_boards;
                prl_launch_fhp_expand(s->fp, region, elem, at, nb, s->d_fill, nullptr, s->d_stage, s->stream);
                PT_TRY(hipGetLastError());
                const long long n = (long long)nb * n_dec * s->R;
                PRL_LAUNCH(prl_k_table_fill, (unsigned)((n + 255) / 256), 256, 0, s->stream, (const void*)s->d_stage, elem, s->R, nb * n_dec, n_dec, s->ncb, m, m + n_dec,
                           m + 2 * n_dec, n_act, (long long)n_plain + (long long)at * n_dec, T->probs, (const int32_t*)nullptr);
                PT_TRY(hipGetLastError());
            }
        }
        PT_TRY(hipStreamSynchronize(s->stream));
    }
#undef PT_TRY
    (void)hipFree(d_meta);
    *out = T;
    return PRL_OK;
}
