// ilupp_amd/csrc/ml_order.hip -- the ORDER decisions of the multilevel preconditioner's preprocessing.
//
//   * The maximum-weight perfect matching with its scalings (what the reference gets from find_pmwm, pmwm_implementation.h:385-537):
//     the assignment problem on c(i,j) = log(max_k |a(i,k)| / |a(i,j)|) solved by shortest augmenting paths from heuristically
//     initialised duals (Duff & Koster, "On algorithms for permuting large entries to the diagonal of a sparse matrix", SIMAX 22, 2001).
//     Here the work is split by what each side is good at:
//       device  the ratios max/|a| (IEEE division), the dual initialisation (column minima by 64-bit atomic minimum on the bit pattern of
//               the non-negative costs, row minima of the reduced costs), and the first matching on tight edges -- sequential by
//               definition ("every row in turn takes its first free tight column") but computed in parallel rounds that provably give
//               the sequential answer (k_mw_want / k_mw_take below);
//       host    log and exp: the costs and the final scalings must be the C library's own roundings (the results are compared bit for
//               bit with a host build of the reference), so they are evaluated by the host's libm -- on all host cores, in place in a
//               pinned staging buffer the DMA engines read and write; and the augmenting-path searches for the rows the first matching
//               left over (class PathSearch; none for a matrix with a dominant diagonal).
//     The matching and the duals depend on the order in which equal candidates are taken (a binary heap keyed by path length only:
//     std::priority_queue, as the reference uses the standard library's heap); the searches therefore run in root order, one at a time.
//   * the diagonally-dominant move-to-corner ordering (sparse_implementation.h:4967-5036), as far as the reference defines it;
//   * the sparse-columns-first ordering (column_perm, pmwm_implementation.h:539-560).
#include <math.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

#include "common.h"

namespace ilupp {

namespace {

// ------------------------------------------------------------ host helpers ------------------------------------------------------------
unsigned host_threads()
{
    static const unsigned t = [] {
        if (const char *e = getenv("ILUPP_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return (unsigned)v; }
        const unsigned hw = std::thread::hardware_concurrency();
        return hw == 0 ? 1u : (hw > 16 ? 16u : hw);
    }();
    return t;
}

// f(lo, hi) over [0, count) in contiguous pieces, one per host thread (the pieces are independent: elementwise libm calls)
template <class F> void host_parallel(size_t count, size_t grain, F f)
{
    size_t pieces = std::min<size_t>(host_threads(), count / (grain > 0 ? grain : 1));
    if (pieces <= 1) { f((size_t)0, count); return; }
    std::vector<std::thread> pool;
    pool.reserve(pieces - 1);
    const size_t step = (count + pieces - 1) / pieces;
    for (size_t k = 1; k < pieces; ++k) {
        const size_t lo = std::min(count, k * step), hi = std::min(count, lo + step);
        pool.emplace_back([=] { f(lo, hi); });
    }
    f((size_t)0, std::min(count, step));
    for (std::thread &t : pool) t.join();
}

// one pinned staging buffer for the whole process (constructions run one at a time, api.hip g_build_mu); grows, never shrinks until
// ilupp_hip_release_cached_memory
struct Stage { void *p = nullptr; size_t bytes = 0; std::mutex mu; } g_stage;      // (mu: the workers of a batched construction take turns)

void *stage(size_t bytes)
{
    if (g_stage.bytes < bytes) {
        if (g_stage.p) { (void)hipHostFree(g_stage.p); g_stage.p = nullptr; g_stage.bytes = 0; }
        const size_t want = bytes + bytes / 4;
        ILUPP_HIP(hipHostMalloc(&g_stage.p, want, hipHostMallocDefault));
        g_stage.bytes = want;
    }
    return g_stage.p;
}

// ---------------------------------------------------------- device: initialisation ----------------------------------------------------------
constexpr unsigned long long kUnset = ~0ull;

// ratio[q] = max_k |a(r,k)| / |a(r,q)| (the argument of the cost's logarithm), rowmax[r]
__global__ void k_mw_ratio(int32_t n, const int32_t *__restrict__ ptr, const double *__restrict__ val, double *__restrict__ rowmax, double *__restrict__ ratio)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int q0 = ptr[r], q1 = ptr[r + 1];
    double m = 0.0;
    for (int q = q0; q < q1; ++q) { const double a = fabs(val[q]); if (m < a) m = a; }
    rowmax[r] = m;
    for (int q = q0; q < q1; ++q) ratio[q] = m / fabs(val[q]);
}

// v[c] = the smallest cost of column c.  The costs are >= +0 (ratio >= 1), so their bit patterns order like the numbers: one 64-bit
// atomic minimum per entry.  A NaN cost (a row of stored zeros) leaves the reference's comparisons order dependent: flagged, and the
// host's sequential initialisation takes over.
__global__ void k_mw_colmin(int64_t nnz, const int32_t *__restrict__ idx, const double *__restrict__ cost, unsigned long long *__restrict__ vbits, int32_t *flag)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const double c = cost[q];
    if (c != c || c < 0.0) { *flag = 1; return; }
    atomicMin(&vbits[idx[q]], (unsigned long long)__double_as_longlong(c));
}

// v as numbers (-1 = "no entry in this column", as the reference marks it); u[r] = the smallest reduced cost of row r (-1 for an empty row)
__global__ void k_mw_rowmin(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ cost,
                            const unsigned long long *__restrict__ vbits, double *__restrict__ u, int32_t *flag)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    double m = -1.0;
    bool first = true;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const double red = cost[q] - __longlong_as_double((long long)vbits[idx[q]]);
        if (red != red) { *flag = 1; return; }                 // inf - inf: an all-zero column under a stored zero
        if (first || m > red) { m = red; first = false; }
    }
    u[r] = m;
}
__global__ void k_mw_v(int32_t n, const unsigned long long *__restrict__ vbits, double *__restrict__ v)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) v[c] = vbits[c] == kUnset ? -1.0 : __longlong_as_double((long long)vbits[c]);
}

// ---- the first matching: "for r = 0, 1, ...: r takes the first column of its row that is tight and still free" -- in parallel rounds.
// A row is UNDECIDED (-1), MATCHED (>= 0) or has run OUT of columns (-2).  In a round every undecided row writes its index to every free
// tight column of its row (minimum wins); then it looks at its first free tight column: if the minimum there is itself, no undecided
// row before it wants that column at all, every column before it in the row is taken for good, so the sequential loop would give it
// exactly this column: it takes it.  Otherwise it waits for the next round.  The smallest undecided row always decides, most rows
// decide in the first round (a row rarely has more than one tight entry).
#define MW_TIGHT(r, q, c) (cost[q] - u[r] - v[c] == 0)        // the reference's test on the reduced cost, evaluated in its order

__global__ void k_mw_want(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ cost,
                          const double *__restrict__ u, const double *__restrict__ v, const int32_t *__restrict__ row_mate,
                          const int32_t *__restrict__ col_mate, int32_t *__restrict__ want)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || row_mate[r] != -1) return;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        if (col_mate[c] == -1 && MW_TIGHT(r, q, c)) atomicMin(&want[c], r);
    }
}
__global__ void k_mw_take(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ idx, const double *__restrict__ cost,
                          const double *__restrict__ u, const double *__restrict__ v, int32_t *__restrict__ row_mate, int32_t *col_mate,
                          int32_t *__restrict__ col_edge, const int32_t *__restrict__ want, int32_t *pending)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n || row_mate[r] != -1) return;
    for (int q = ptr[r]; q < ptr[r + 1]; ++q) {
        const int c = idx[q];
        // (a column another row takes in this very launch reads as free or as taken: free -> its minimum is that row, this one waits;
        //  taken -> this row moves on to a column whose minimum was computed with both of them competing.  Same outcome either way.)
        if (__hip_atomic_load(&col_mate[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1 || !MW_TIGHT(r, q, c)) continue;
        if (want[c] == r) {
            row_mate[r] = c;
            __hip_atomic_store(&col_mate[c], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            col_edge[c] = q;
        } else *pending = 1;
        return;
    }
    row_mate[r] = -2;
}
// rows that ran out of columns are plain unmatched rows from here on; count them
__global__ void k_mw_leftover(int32_t n, int32_t *__restrict__ row_mate, int32_t *count)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (row_mate[r] < 0) { row_mate[r] = -1; atomicAdd(count, 1); }
}

__global__ void k_mw_identity(int32_t n, int32_t *__restrict__ p, double *__restrict__ d1, double *__restrict__ d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { p[i] = i; d1[i] = 1.0; d2[i] = 1.0; }
}

// ------------------------------------------------------ host: what the first matching left ------------------------------------------------------
// The state of the assignment problem on the host: costs, duals, the matching (row_mate / col_mate, -1 = free) and for every matched
// column the matrix entry that matches it.
struct Assignment {
    int32_t n;
    const int32_t *ptr, *idx;
    const double *cost;
    double *u, *v;
    int32_t *row_mate, *col_mate, *col_edge;
    bool tight(int32_t r, int32_t q) const { return cost[q] - u[r] - v[idx[q]] == 0; }
    void match(int32_t r, int32_t q) { const int32_t c = idx[q]; row_mate[r] = c; col_mate[c] = r; col_edge[c] = q; }
};

// the sequential form of the initialisation (used when the device's rounds do not apply: NaN costs, or a pathological chain of rows
// waiting for one another)
void sequential_start(Assignment &S)
{
    const int32_t n = S.n;
    for (int32_t i = 0; i < n; ++i) { S.v[i] = -1; S.u[i] = -1; S.row_mate[i] = -1; S.col_mate[i] = -1; }
    for (int32_t q = 0; q < S.ptr[n]; ++q) { double &vc = S.v[S.idx[q]]; if (vc > S.cost[q] || vc == -1) vc = S.cost[q]; }
    for (int32_t r = 0; r < n; ++r)
        for (int32_t q = S.ptr[r]; q < S.ptr[r + 1]; ++q) {
            const double red = S.cost[q] - S.v[S.idx[q]];
            if (S.u[r] > red || S.u[r] == -1) S.u[r] = red;
        }
    for (int32_t r = 0; r < n; ++r)
        for (int32_t q = S.ptr[r]; q < S.ptr[r + 1]; ++q)
            if (S.col_mate[S.idx[q]] == -1 && S.tight(r, q)) { S.match(r, q); break; }
}

// a free row next to a matched column whose row has a free tight column of its own: both get matched (an augmenting path of two
// tight edges costs nothing and needs no search)
void two_edge_paths(Assignment &S)
{
    for (int32_t r = 0; r < S.n; ++r) {
        if (S.row_mate[r] != -1) continue;
        for (int32_t q = S.ptr[r]; q < S.ptr[r + 1] && S.row_mate[r] == -1; ++q) {
            const int32_t c = S.idx[q];
            if (S.col_mate[c] == -1 || !S.tight(r, q)) continue;
            const int32_t other = S.col_mate[c];
            for (int32_t q2 = S.ptr[other]; q2 < S.ptr[other + 1]; ++q2)
                if (S.col_mate[S.idx[q2]] == -1 && S.tight(other, q2)) { S.match(other, q2); S.match(r, q); break; }
        }
    }
}

// Shortest augmenting paths in the graph of reduced costs, one root at a time (Dijkstra over columns; a column is reached through a
// row, continues through its matched row).  Per-search state is kept in dense arrays validated by a search number.
class PathSearch {
    struct Reached { double dist; int32_t col; bool operator>(const Reached &o) const { return dist > o.dist; } };
    Assignment &S;
    std::vector<uint32_t> seen, done;        // == number of the current search: the column has a distance / is settled
    std::vector<double> dist;
    std::vector<int32_t> via;                // the entry (q) through which the column got its current distance
    std::vector<int32_t> from;               // for a matched row: the row its column was reached from
    std::vector<int32_t> settled;            // the settled columns of the current search
    uint32_t search = 0;

public:
    explicit PathSearch(Assignment &s) : S(s), seen((size_t)s.n, 0), done((size_t)s.n, 0), dist((size_t)s.n, 0.0), via((size_t)s.n, -1), from((size_t)s.n, -1) {}

    // false: no free column can be reached from `root` (no perfect matching)
    bool augment_from(int32_t root)
    {
        ++search;
        settled.clear();
        std::priority_queue<Reached, std::vector<Reached>, std::greater<Reached>> frontier;
        double shortest = -1, here = 0;       // shortest: length of the best path to a free column so far (-1: none)
        int32_t last_row = -1, free_col = -1;
        for (int32_t r = root;;) {
            for (int32_t q = S.ptr[r]; q < S.ptr[r + 1]; ++q) {
                const int32_t c = S.idx[q];
                if (done[(size_t)c] == search) continue;
                const double d = here + S.cost[q] - S.u[r] - S.v[c];
                if (!(shortest == -1 || d < shortest)) continue;
                if (S.col_mate[c] == -1) { shortest = d; via[(size_t)c] = q; free_col = c; last_row = r; }
                else if (seen[(size_t)c] != search || d < dist[(size_t)c]) {
                    seen[(size_t)c] = search; dist[(size_t)c] = d; via[(size_t)c] = q;
                    from[(size_t)S.col_mate[c]] = r;
                    frontier.push(Reached{d, c});
                }
            }
            bool have = false;
            Reached next{0.0, -1};
            while (!frontier.empty()) {
                next = frontier.top(); frontier.pop();
                if (done[(size_t)next.col] != search) { have = true; break; }     // (an entry of a column settled since it was pushed: stale)
            }
            if (!have) break;
            here = next.dist;
            if (shortest != -1 && shortest <= here) break;
            done[(size_t)next.col] = search;
            settled.push_back(next.col);
            r = S.col_mate[next.col];
        }
        if (shortest == -1 || free_col == -1) return false;
        // flip the path: every row on it takes the column it was reached... through which the path came to it
        for (int32_t r = last_row, c = free_col;;) {
            const int32_t released = S.row_mate[r];
            S.match(r, via[(size_t)c]);
            if (r == root) break;
            c = released;
            r = from[(size_t)r];
        }
        // the duals: settled columns move by their distance to the end of the path; every row matched to a settled column, and the row
        // now matched to the column the path ended in, become tight on their matching entries again
        for (int32_t c : settled) S.v[c] = S.v[c] + dist[(size_t)c] - shortest;
        for (int32_t c : settled) S.u[S.col_mate[c]] = S.cost[S.col_edge[c]] - S.v[c];
        S.u[S.col_mate[free_col]] = S.cost[S.col_edge[free_col]] - S.v[free_col];
        return true;
    }
};

}  // namespace

void mwm_release_stage()
{
    std::lock_guard<std::mutex> stage_turn(g_stage.mu);
    if (g_stage.p) (void)hipHostFree(g_stage.p);
    g_stage.p = nullptr; g_stage.bytes = 0;
}

// The maximum-weight matching of A (ROW storage, on the device) and its scalings, left on the device:
// p1[c] = the row matched to column c; D1, D2: the reciprocal row / column scalings of matrix_sparse::preprocess :5276-5279
// (D1[r] = max_k |a(r,k)| / exp(u[r]), D2[c] = exp(-v[c])).  Without a perfect matching: identity and ones (:460-471).
int mwm_order(hipStream_t st, const DevMat &A, int32_t *p1, double *D1, double *D2)
{
    const int32_t n = A.n;
    const int64_t nnz = A.nnz;
    const int gb = (n + 255) / 256;
    const bool dbg = getenv("ILUPP_DEBUG") != nullptr;
    auto identity = [&]() { hipLaunchKernelGGL(k_mw_identity, dim3(gb), dim3(256), 0, st, n, p1, D1, D2); return ILUPP_OK; };
    if (n <= 0) return ILUPP_OK;
    if (nnz <= 0) return identity();
    PoolBlock b_cost, b_rowmax, b_vbits, b_u, b_v, b_rm, b_ce, b_want, b_flag;
    ILUPP_HIP(b_cost.alloc(sizeof(double) * (size_t)nnz));
    for (PoolBlock *b : {&b_rowmax, &b_vbits, &b_u, &b_v}) ILUPP_HIP(b->alloc(sizeof(double) * (size_t)n));
    for (PoolBlock *b : {&b_rm, &b_ce, &b_want}) ILUPP_HIP(b->alloc(sizeof(int32_t) * (size_t)n));
    ILUPP_HIP(b_flag.alloc(64));
    double *cost = b_cost.as<double>(), *rowmax = b_rowmax.as<double>(), *u = b_u.as<double>(), *v = b_v.as<double>();
    unsigned long long *vbits = b_vbits.as<unsigned long long>();
    int32_t *row_mate = b_rm.as<int32_t>(), *col_mate = p1, *col_edge = b_ce.as<int32_t>(), *want = b_want.as<int32_t>(), *flag = b_flag.as<int32_t>();
    std::lock_guard<std::mutex> stage_turn(g_stage.mu);
    // host side of the stage: costs (nnz), then u, v, rowmax (n each), then row_mate, col_mate, col_edge (n each)
    char *hs = static_cast<char *>(stage(sizeof(double) * ((size_t)nnz + 3 * (size_t)n) + sizeof(int32_t) * 3 * (size_t)n));
    double *h_cost = reinterpret_cast<double *>(hs), *h_u = h_cost + nnz, *h_v = h_u + n, *h_rowmax = h_v + n;
    int32_t *h_rm = reinterpret_cast<int32_t *>(h_rowmax + n), *h_cm = h_rm + n, *h_ce = h_cm + n;

    // 1. ratios on the device, logarithms on the host's cores, costs back
    hipLaunchKernelGGL(k_mw_ratio, dim3(gb), dim3(256), 0, st, n, A.ptr, A.val, rowmax, cost);
    ILUPP_HIP(hipMemcpyAsync(h_cost, cost, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost, st));
    ILUPP_HIP(hipStreamSynchronize(st));
    host_parallel((size_t)nnz, 1u << 14, [=](size_t lo, size_t hi) { for (size_t q = lo; q < hi; ++q) h_cost[q] = log(h_cost[q]); });
    ILUPP_HIP(hipMemcpyAsync(cost, h_cost, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, st));

    // 2. duals and the first matching on the device
    ILUPP_HIP(hipMemsetAsync(vbits, 0xff, sizeof(double) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(flag, 0, 64, st));
    ILUPP_HIP(hipMemsetAsync(row_mate, 0xff, sizeof(int32_t) * (size_t)n, st));
    ILUPP_HIP(hipMemsetAsync(col_mate, 0xff, sizeof(int32_t) * (size_t)n, st));
    hipLaunchKernelGGL(k_mw_colmin, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, A.idx, (const double *)cost, vbits, flag);
    hipLaunchKernelGGL(k_mw_rowmin, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, (const double *)cost, (const unsigned long long *)vbits, u, flag);
    hipLaunchKernelGGL(k_mw_v, dim3(gb), dim3(256), 0, st, n, (const unsigned long long *)vbits, v);
    int32_t h_flag[4] = {0, 0, 0, 0};                              // [0] NaN seen, [1] a row waits, [2] rows left unmatched
    bool on_device = true;
    int rounds = 0;
    for (;; ++rounds) {
        ILUPP_HIP(hipMemsetAsync(want, 0x7f, sizeof(int32_t) * (size_t)n, st));
        ILUPP_HIP(hipMemsetAsync(flag + 1, 0, sizeof(int32_t), st));
        hipLaunchKernelGGL(k_mw_want, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, (const double *)cost, (const double *)u, (const double *)v,
                           (const int32_t *)row_mate, (const int32_t *)col_mate, want);
        hipLaunchKernelGGL(k_mw_take, dim3(gb), dim3(256), 0, st, n, A.ptr, A.idx, (const double *)cost, (const double *)u, (const double *)v, row_mate,
                           col_mate, col_edge, (const int32_t *)want, flag + 1);
        ILUPP_HIP(hipMemcpyAsync(h_flag, flag, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
        if (h_flag[0] || (h_flag[1] && rounds >= 48)) { on_device = false; break; }
        if (!h_flag[1]) break;
    }
    int32_t left = 0;
    if (on_device) {
        hipLaunchKernelGGL(k_mw_leftover, dim3(gb), dim3(256), 0, st, n, row_mate, flag + 2);
        ILUPP_HIP(hipMemcpyAsync(&left, flag + 2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
    }
    if (dbg) fprintf(stderr, "[ilupp] ml: matching: %d rounds on the device%s, %d rows left for the path search\n", rounds + 1,
                     on_device ? "" : " abandoned (sequential start on the host)", on_device ? left : -1);

    // 3. whatever is left: the searches on the host (the structure of A comes over only now)
    ILUPP_HIP(hipMemcpyAsync(h_rowmax, rowmax, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    if (!on_device || left > 0) {
        std::vector<int32_t> hp((size_t)n + 1), hi((size_t)nnz);
        ILUPP_HIP(hipMemcpyAsync(hp.data(), A.ptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipMemcpyAsync(hi.data(), A.idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, st));
        if (on_device) {
            ILUPP_HIP(hipMemcpyAsync(h_u, u, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipMemcpyAsync(h_v, v, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipMemcpyAsync(h_rm, row_mate, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipMemcpyAsync(h_cm, col_mate, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
            ILUPP_HIP(hipMemcpyAsync(h_ce, col_edge, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
        }
        ILUPP_HIP(hipStreamSynchronize(st));
        Assignment S{n, hp.data(), hi.data(), h_cost, h_u, h_v, h_rm, h_cm, h_ce};
        if (!on_device) sequential_start(S);
        two_edge_paths(S);
        PathSearch paths(S);
        for (int32_t root = 0; root < n; ++root)
            if (S.row_mate[root] == -1 && !paths.augment_from(root)) return identity();
        ILUPP_HIP(hipMemcpyAsync(col_mate, h_cm, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
    } else {
        ILUPP_HIP(hipMemcpyAsync(h_u, u, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipMemcpyAsync(h_v, v, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
        ILUPP_HIP(hipStreamSynchronize(st));
    }

    // 4. the scalings: exp on the host's cores, in place
    host_parallel((size_t)n, 1u << 13, [=](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) { h_u[i] = h_rowmax[i] / exp(h_u[i]); h_v[i] = exp(-h_v[i]); }
    });
    ILUPP_HIP(hipMemcpyAsync(D1, h_u, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    ILUPP_HIP(hipMemcpyAsync(D2, h_v, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    ILUPP_HIP(hipStreamSynchronize(st));                       // (the stage is free for the next caller)
    return ILUPP_OK;
}

// The ordering that grows a diagonally dominant leading block (sparse_implementation.h:4967-5012): repeatedly the index with the smallest
// accumulated weight towards the block; accepted while every row and column of the block keeps its off-diagonal mass <= 2.  `tptr/tidx/tval`:
// the column-major copy.  Returns false as soon as an index is rejected: the reference then refills its container through resize()
// (:5014) with the `used` flags of the first phase still set (arrays_implementation.h:55-63), after which its result is not a permutation.
bool dd_move_corner_host(int32_t n, const int32_t *ptr, const int32_t *idx, const double *val, const int32_t *tptr, const int32_t *tidx, const double *tval,
                         std::vector<int32_t> &P)
{
    typedef std::multimap<double, int32_t> Pool;
    Pool pool;
    std::vector<Pool::iterator> at((size_t)n);
    std::vector<char> in_pool((size_t)n, 1), in_block((size_t)n, 0);
    std::vector<double> row_gap((size_t)n, 2.0), col_gap((size_t)n, 2.0);
    for (int32_t k = 0; k < n; ++k) at[(size_t)k] = pool.insert(Pool::value_type(0.0, k));
    auto add = [&](int32_t k, double w) {
        double old = 0.0;
        if (in_pool[(size_t)k]) { old = at[(size_t)k]->first; pool.erase(at[(size_t)k]); }
        at[(size_t)k] = pool.insert(Pool::value_type(w + old, k));
        in_pool[(size_t)k] = 1;
    };
    P.assign((size_t)n, 0);
    for (int32_t step = 0; step < n; ++step) {
        const int32_t cur = pool.begin()->second;
        bool ok = row_gap[(size_t)cur] >= 0 && col_gap[(size_t)cur] >= 0;
        for (int32_t q = ptr[cur]; ok && q < ptr[cur + 1]; ++q) if (in_block[(size_t)idx[q]]) ok = fabs(val[q]) <= col_gap[(size_t)idx[q]];
        for (int32_t q = tptr[cur]; ok && q < tptr[cur + 1]; ++q) if (in_block[(size_t)tidx[q]]) ok = fabs(tval[q]) <= row_gap[(size_t)tidx[q]];
        if (!ok) return false;
        P[(size_t)step] = cur;
        in_pool[(size_t)cur] = 0;
        pool.erase(pool.begin());
        in_block[(size_t)cur] = 1;
        for (int32_t q = ptr[cur]; q < ptr[cur + 1]; ++q) {
            const int32_t c = idx[q];
            if (!in_block[(size_t)c]) add(c, fabs(val[q]));
            col_gap[(size_t)c] -= fabs(val[q]);
        }
        for (int32_t q = tptr[cur]; q < tptr[cur + 1]; ++q) {
            const int32_t c = tidx[q];
            if (!in_block[(size_t)c]) add(c, fabs(tval[q]));
            row_gap[(size_t)c] -= fabs(tval[q]);
        }
    }
    return true;
}

// columns by increasing number of entries (column_perm, pmwm_implementation.h:539-560: the reference's quicksort on the counts; identity
// when a column is empty)
void sparse_first_host(int32_t n, std::vector<int32_t> &counts, std::vector<int32_t> &p2)
{
    p2.resize((size_t)n);
    for (int32_t k = 0; k < n; ++k) p2[(size_t)k] = k;
    if (n > 0) ref_quicksort(counts.data(), p2.data(), 0, (long)n - 1);
    if (n > 0 && counts[0] == 0) for (int32_t k = 0; k < n; ++k) p2[(size_t)k] = k;
}

}  // namespace ilupp
