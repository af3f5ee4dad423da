// ilupp_amd/csrc/ml_order.hip -- the ORDER decisions of the multilevel preconditioner's preprocessing that are sequential algorithms by
// definition and run on the host (host code only; SURVEY section 8f rank 3 allows the matching on the host in the first cut):
//   * the maximum-weight perfect matching with its scalings (reference find_pmwm, pmwm_implementation.h:385-537, class sapTree :37-383):
//     Duff-Koster style shortest augmenting paths on c(i,j) = log(max_k |a(i,k)| / |a(i,j)|) with heuristically initialised duals;
//   * the diagonally-dominant move-to-corner ordering (sparse_implementation.h:4967-5036), as far as the reference defines it;
//   * the sparse-columns-first ordering (column_perm, pmwm_implementation.h:539-560).
// The results depend on the order in which equal candidates are taken (a binary heap of path lengths, a multimap of weights, an unstable
// quicksort of counts); libstdc++'s own containers are used where the reference uses them, so equal candidates come out in its order.
#include <math.h>

#include <map>
#include <queue>
#include <vector>

#include "common.h"

namespace ilupp {

namespace {

// a set of column indices with values, listing its members in insertion order (what the reference keeps in vector_sparse_dynamic objects:
// membership = "has a slot", sparse.h:191; a member whose value is set to 0 stays a member)
struct IndexedSet {
    std::vector<int32_t> where, members;
    std::vector<double> value;
    explicit IndexedSet(int32_t n) : where((size_t)n, -1) {}
    bool has(int32_t j) const { return where[(size_t)j] >= 0; }
    double &at(int32_t j)
    {
        if (where[(size_t)j] < 0) { where[(size_t)j] = (int32_t)members.size(); members.push_back(j); value.push_back(0.0); }
        return value[(size_t)where[(size_t)j]];
    }
    void clear() { for (int32_t j : members) where[(size_t)j] = -1; members.clear(); value.clear(); }
};

struct Cand {
    int32_t col; double dist, weight;
    bool operator>(const Cand &o) const { return dist > o.dist; }
};

}  // namespace

// mate_col[c] = the row matched to column c; inv_row / inv_col: the reciprocal scalings (D1, D2 of matrix_sparse::preprocess :5276-5279).
// Without a perfect matching: identity and ones (:460-471).
bool pmwm_host(int32_t n, const int32_t *ptr, const int32_t *idx, const double *val, std::vector<int32_t> &mate_col, std::vector<double> &inv_row,
               std::vector<double> &inv_col)
{
    const size_t nz = (size_t)ptr[n];
    std::vector<double> u((size_t)n), v((size_t)n), cost(nz), rowmax((size_t)n, 0.0), cand_weight((size_t)n, 0.0), weight_of((size_t)n, 0.0);
    std::vector<int32_t> mate_row((size_t)n, -1), up((size_t)n, 0);
    mate_col.assign((size_t)n, -1);
    inv_row.assign((size_t)n, 0.0); inv_col.assign((size_t)n, 0.0);
    for (int32_t r = 0; r < n; ++r)
        for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) if (rowmax[(size_t)r] < fabs(val[q])) rowmax[(size_t)r] = fabs(val[q]);
    for (int32_t r = 0; r < n; ++r)
        for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) cost[(size_t)q] = log(rowmax[(size_t)r] / fabs(val[q]));
    // duals: column minima, then row minima of the reduced costs (:171-200; -1 marks "not set")
    for (int32_t i = 0; i < n; ++i) { v[(size_t)i] = -1; u[(size_t)i] = -1; }
    for (size_t q = 0; q < nz; ++q) { const size_t c = (size_t)idx[q]; if (v[c] > cost[q] || v[c] == -1) v[c] = cost[q]; }
    for (int32_t r = 0; r < n; ++r)
        for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) {
            const double red = cost[(size_t)q] - v[(size_t)idx[q]];
            if (u[(size_t)r] > red || u[(size_t)r] == -1) u[(size_t)r] = red;
        }
    auto tight = [&](int32_t r, int32_t q) { return cost[(size_t)q] - u[(size_t)r] - v[(size_t)idx[q]] == 0; };
    // a first matching on tight edges, then paths of length two (:203-250)
    for (int32_t r = 0; r < n; ++r)
        for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) {
            const int32_t c = idx[q];
            if (mate_col[(size_t)c] == -1 && tight(r, q)) { mate_row[(size_t)r] = c; mate_col[(size_t)c] = r; weight_of[(size_t)c] = cost[(size_t)q]; break; }
        }
    for (int32_t r = 0; r < n; ++r) {
        if (mate_row[(size_t)r] != -1) continue;
        for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) {
            const int32_t c = idx[q];
            if (mate_col[(size_t)c] != -1 && tight(r, q)) {
                const int32_t r2 = mate_col[(size_t)c];
                for (int32_t q2 = ptr[r2]; q2 < ptr[r2 + 1]; ++q2) {
                    const int32_t c2 = idx[q2];
                    if (mate_col[(size_t)c2] == -1 && tight(r2, q2)) {
                        mate_row[(size_t)r] = c; mate_col[(size_t)c] = r;
                        mate_row[(size_t)r2] = c2; mate_col[(size_t)c2] = r2;
                        weight_of[(size_t)c2] = cost[(size_t)q2]; weight_of[(size_t)c] = cost[(size_t)q];
                        break;
                    }
                }
            }
            if (mate_row[(size_t)r] != -1) break;
        }
    }
    IndexedSet settled(n), reached(n);
    for (int32_t root = 0; root < n; ++root) {
        if (mate_row[(size_t)root] != -1) continue;
        std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> heap;
        settled.clear(); reached.clear();
        // shortest augmenting path from `root` (:287-355)
        double best = -1, base = 0;
        int32_t end_row = -1, end_col = -1, r = root;
        for (;;) {
            for (int32_t q = ptr[r]; q < ptr[r + 1]; ++q) {
                const int32_t c = idx[q];
                if (settled.has(c)) continue;
                const double w = cost[(size_t)q];
                const double d = base + w - u[(size_t)r] - v[(size_t)c];
                if (best == -1 || d < best) {
                    if (mate_col[(size_t)c] == -1) { best = d; cand_weight[(size_t)c] = w; end_col = c; end_row = r; }
                    else if (!reached.has(c) || d < reached.at(c)) {
                        reached.at(c) = d;
                        up[(size_t)mate_col[(size_t)c]] = r;
                        heap.push(Cand{c, d, w});
                    }
                }
            }
            if (heap.empty()) break;
            Cand top;
            do { top = heap.top(); heap.pop(); } while (settled.has(top.col) && !heap.empty());
            if (heap.empty() && settled.has(top.col)) break;
            base = top.dist;
            if (best != -1 && best <= base) break;
            cand_weight[(size_t)top.col] = top.weight;
            settled.at(top.col) = 1;
            r = mate_col[(size_t)top.col];
        }
        if (best == -1 || end_col == -1) {
            for (int32_t s = 0; s < n; ++s) { inv_row[(size_t)s] = 1.0; inv_col[(size_t)s] = 1.0; mate_col[(size_t)s] = s; }
            return false;
        }
        // augment along the path that ends in (end_row, end_col) (:144-168)
        {
            int32_t i = end_row, j = end_col;
            mate_col[(size_t)j] = i;
            while (i != root) {
                weight_of[(size_t)j] = cand_weight[(size_t)j];
                const int32_t k = mate_row[(size_t)i];
                mate_row[(size_t)i] = j;
                j = k;
                i = up[(size_t)i];
                mate_col[(size_t)j] = i;
            }
            mate_row[(size_t)root] = j;
            weight_of[(size_t)j] = cand_weight[(size_t)j];
        }
        // the duals (:253-284): the settled columns first, then the rows of the path, then the rows matched to the settled columns
        for (size_t s = 0; s < settled.members.size(); ++s) { const int32_t c = settled.members[s]; v[(size_t)c] = v[(size_t)c] + reached.at(c) - best; }
        {
            int32_t i = end_row, j = end_col;
            while (i != root) {
                settled.at(j) = 0;
                u[(size_t)i] = weight_of[(size_t)j] - v[(size_t)j];
                i = up[(size_t)i]; j = mate_row[(size_t)i];
            }
            settled.at(j) = 0;
            u[(size_t)root] = weight_of[(size_t)j] - v[(size_t)j];
        }
        for (size_t s = 0; s < settled.members.size(); ++s) { const int32_t c = settled.members[s]; u[(size_t)mate_col[(size_t)c]] = weight_of[(size_t)c] - v[(size_t)c]; }
    }
    for (int32_t s = 0; s < n; ++s) { inv_row[(size_t)s] = rowmax[(size_t)s] / exp(u[(size_t)s]); inv_col[(size_t)s] = exp(-v[(size_t)s]); }
    return true;
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
static void count_quicksort(int32_t *data, int32_t *list, long left, long right)
{
    while (left < right) {
        const int32_t m = data[left];
        long i = left, j = right;
        while (i <= j) {
            while (data[i] < m) i++;
            while (data[j] > m) j--;
            if (i <= j) {
                const int32_t t = data[i]; data[i] = data[j]; data[j] = t;
                const int32_t w = list[i]; list[i] = list[j]; list[j] = w;
                i++; j--;
            }
        }
        if (j - left < right - i) { count_quicksort(data, list, left, j); left = i; }
        else { count_quicksort(data, list, i, right); right = j; }
    }
}
void sparse_first_host(int32_t n, std::vector<int32_t> &counts, std::vector<int32_t> &p2)
{
    p2.resize((size_t)n);
    for (int32_t k = 0; k < n; ++k) p2[(size_t)k] = k;
    if (n > 0) count_quicksort(counts.data(), p2.data(), 0, (long)n - 1);
    if (n > 0 && counts[0] == 0) for (int32_t k = 0; k < n; ++k) p2[(size_t)k] = k;
}

}  // namespace ilupp
