/*
 * oracle/ilupp_oracle.c -- CPU restatement of the ilupp hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain C99, single-threaded, written from the behavioural spec of the reference (file:line cited
 * per function, paths relative to /root/reference/src/ilupp unless noted).  Operation ORDER follows
 * the reference exactly so that results are bit-identical on x86-64 without FMA contraction
 * (compile with -O2 -ffp-contract=off; see oracle/Makefile).
 *
 * Parity PINNED by tests/test_oracle_golden.py against golden vectors emitted by the reference
 * itself (tests/golden/make_golden.py via oracle/_ref).
 *
 * Nothing in the product path (ilupp_amd/, include/) may link, load or call this file.
 */
#include "ilupp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

static void mat_init(orc_mat *M, orc_int n, orc_int cap, int is_csr)
{
    M->n = n;
    M->nnz = 0;
    M->is_csr = is_csr;
    M->ptr = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));   /* reformat zero-fills pointer (:2531-2540) */
    M->idx = (orc_int *)malloc(sizeof(orc_int) * (size_t)(cap > 0 ? cap : 1));
    M->val = (double *)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
}

void orc_free_mat(orc_mat *M)
{
    if (!M) return;
    free(M->ptr); free(M->idx); free(M->val);
    M->ptr = M->idx = NULL; M->val = NULL; M->n = M->nnz = 0;
}

static void mat_swap(orc_mat *A, orc_mat *B)   /* matrix_sparse::interchange, sparse_implementation.h:3475-3484 */
{
    orc_mat t = *A; *A = *B; *B = t;
}

/* matrix_sparse::compress(threshold), sparse_implementation.h:3696-3722: keep |x| > thr, exact size. */
static void mat_compress(orc_mat *M, double thr)
{
    orc_int counter = 0, i, j, start = 0;
    for (i = 0; i < M->n; ++i) {
        orc_int end = M->ptr[i + 1];
        for (j = start; j < end; ++j) {
            if (fabs(M->val[j]) > thr) {
                M->val[counter] = M->val[j];
                M->idx[counter] = M->idx[j];
                counter++;
            }
        }
        start = end;
        M->ptr[i + 1] = counter;
    }
    M->nnz = counter;
    M->idx = (orc_int *)realloc(M->idx, sizeof(orc_int) * (size_t)(counter > 0 ? counter : 1));
    M->val = (double *)realloc(M->val, sizeof(double) * (size_t)(counter > 0 ? counter : 1));
}

/* ------------------------------------------------------------------------------------------ */
/* ILU(0)                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* sparse_vec_update, ILU0.hpp:8-23: two-pointer merge, update only above column k. */
static void sparse_vec_update(orc_int l1, orc_int u1, orc_int l2, orc_int u2, orc_int k, double l_ik,
                              const orc_int *indices, double *data_U)
{
    while (l1 < u1 && l2 < u2) {
        if (indices[l1] == indices[l2]) {
            if (indices[l1] > k)
                data_U[l1] -= l_ik * data_U[l2];
            l1++; l2++;
        } else if (indices[l1] < indices[l2])
            l1++;
        else
            l2++;
    }
}

int orc_ilu0(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_mat *L, orc_mat *U)
{
    /* compute_ilu0, ILU0.hpp:26-66 */
    const orc_int nnz = ptr[n];
    double *LU = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    double *diag_U = (double *)malloc(sizeof(double) * (size_t)n);
    orc_int diag_idx = 0, nnzL = 0, nnzU = 0, i, kk;

    for (i = 0; i < n; ++i) {
        for (kk = ptr[i]; kk < ptr[i + 1]; ++kk) {
            LU[kk] = val[kk];
            if (idx[kk] == i) diag_idx = kk;
            if (idx[kk] <= i) ++nnzL;
            if (idx[kk] >= i) ++nnzU;
        }
        for (kk = ptr[i]; kk < ptr[i + 1]; ++kk) {
            const orc_int k = idx[kk];
            double L_ik;
            if (k >= i) continue;
            L_ik = LU[kk] / diag_U[k];
            sparse_vec_update(ptr[i], ptr[i + 1], ptr[k], ptr[k + 1], k, L_ik, idx, LU);
            LU[kk] = L_ik;
        }
        diag_U[i] = LU[diag_idx];
    }

    /* split, ILU0.hpp:79-98: L = strict lower + unit diagonal LAST; U = diagonal first + upper */
    mat_init(L, n, nnzL, 1);
    mat_init(U, n, nnzU, 1);
    {
        orc_int iL = 0, iU = 0, k;
        for (i = 0; i < n; ++i) {
            for (k = ptr[i]; k < ptr[i + 1]; ++k) {
                const orc_int j = idx[k];
                if (j < i) { L->idx[iL] = j; L->val[iL++] = LU[k]; }
                else       { U->idx[iU] = j; U->val[iU++] = LU[k]; }
            }
            L->idx[iL] = i; L->val[iL++] = 1.0;
            L->ptr[i + 1] = iL;
            U->ptr[i + 1] = iU;
        }
        L->nnz = iL; U->nnz = iU;
    }
    free(LU); free(diag_U);

    if (!is_csr) {   /* ILU0.hpp:100-105: we decomposed A^T; swap roles and relabel */
        mat_swap(L, U);
        L->is_csr = 0;
        U->is_csr = 0;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* std::sort as shipped by libstdc++ (bits/stl_algo.h:1855-1957, bits/stl_heap.h), restated    */
/* for "slot ids ordered by DEcreasing |key|" -- the comparator of dropping.hpp:25-26.         */
/* Needed because the kept set under magnitude ties is defined by this exact algorithm.        */
/* ------------------------------------------------------------------------------------------ */

typedef struct { const double *key; } abs_desc_cmp;
#define CMP(a, b) (fabs(c->key[(a)]) > fabs(c->key[(b)]))

static void s_unguarded_linear_insert(orc_int *last, const abs_desc_cmp *c)
{
    orc_int v = *last;
    orc_int *next = last - 1;
    while (CMP(v, *next)) { *last = *next; last = next; --next; }
    *last = v;
}

static void s_insertion_sort(orc_int *first, orc_int *last, const abs_desc_cmp *c)
{
    orc_int *i;
    if (first == last) return;
    for (i = first + 1; i != last; ++i) {
        if (CMP(*i, *first)) {
            orc_int v = *i;
            memmove(first + 1, first, sizeof(orc_int) * (size_t)(i - first));
            *first = v;
        } else
            s_unguarded_linear_insert(i, c);
    }
}

static void s_push_heap(orc_int *first, long hole, long top, orc_int v, const abs_desc_cmp *c)
{
    long parent = (hole - 1) / 2;
    while (hole > top && CMP(first[parent], v)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = v;
}

static void s_adjust_heap(orc_int *first, long hole, long len, orc_int v, const abs_desc_cmp *c)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (CMP(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    s_push_heap(first, hole, top, v, c);
}

static void s_heapsort(orc_int *first, orc_int *last, const abs_desc_cmp *c)
{
    /* __partial_sort(first, last, last): __heap_select == __make_heap here, then __sort_heap */
    long len = last - first, parent;
    if (len >= 2) {
        parent = (len - 2) / 2;
        for (;;) {
            orc_int v = first[parent];
            s_adjust_heap(first, parent, len, v, c);
            if (parent == 0) break;
            parent--;
        }
    }
    while (last - first > 1) {
        orc_int v;
        --last;
        v = *last;
        *last = *first;
        s_adjust_heap(first, 0, last - first, v, c);
    }
}

static void s_move_median_to_first(orc_int *result, orc_int *a, orc_int *b, orc_int *cc, const abs_desc_cmp *c)
{
#define SWAP(p, q) do { orc_int t_ = *(p); *(p) = *(q); *(q) = t_; } while (0)
    if (CMP(*a, *b)) {
        if (CMP(*b, *cc)) SWAP(result, b);
        else if (CMP(*a, *cc)) SWAP(result, cc);
        else SWAP(result, a);
    } else if (CMP(*a, *cc)) SWAP(result, a);
    else if (CMP(*b, *cc)) SWAP(result, cc);
    else SWAP(result, b);
}

static orc_int *s_unguarded_partition(orc_int *first, orc_int *last, orc_int *pivot, const abs_desc_cmp *c)
{
    for (;;) {
        while (CMP(*first, *pivot)) ++first;
        --last;
        while (CMP(*pivot, *last)) --last;
        if (!(first < last)) return first;
        SWAP(first, last);
        ++first;
    }
}

static void s_introsort_loop(orc_int *first, orc_int *last, long depth, const abs_desc_cmp *c)
{
    while (last - first > 16) {
        orc_int *mid, *cut;
        if (depth == 0) { s_heapsort(first, last, c); return; }
        --depth;
        mid = first + (last - first) / 2;
        s_move_median_to_first(first, first + 1, mid, last - 1, c);
        cut = s_unguarded_partition(first + 1, last, first, c);
        s_introsort_loop(cut, last, depth, c);
        last = cut;
    }
}

void orc_sort_slots_by_abs_desc(orc_int *list, orc_int len, const double *key)
{
    abs_desc_cmp cmp, *c = &cmp;
    orc_int *first = list, *last = list + len;
    long lg = 0, m = len;
    cmp.key = key;
    if (len <= 0) return;
    while (m > 1) { m >>= 1; ++lg; }          /* std::__lg */
    s_introsort_loop(first, last, 2 * lg, c);
    if (last - first > 16) {                  /* __final_insertion_sort */
        orc_int *i;
        s_insertion_sort(first, first + 16, c);
        for (i = first + 16; i != last; ++i) s_unguarded_linear_insert(i, c);
    } else
        s_insertion_sort(first, last, c);
}
#undef CMP
#undef SWAP

static int cmp_slot_by_index(const void *a, const void *b, void *ctx)
{
    const orc_int *wp = (const orc_int *)ctx;
    orc_int ia = wp[*(const orc_int *)a], ib = wp[*(const orc_int *)b];
    return (ia > ib) - (ia < ib);
}

/* sort kept slots by column index; keys are unique among kept slots (dead slots carry 0 and are
 * never kept), so any correct sort reproduces dropping.hpp:32-33. */
static void sort_slots_by_index(orc_int *list, orc_int len, const orc_int *wpointer)
{
    orc_int i;
    /* plain insertion sort for short lists, qsort_r otherwise */
    if (len <= 32) {
        for (i = 1; i < len; ++i) {
            orc_int v = list[i], j = i - 1;
            while (j >= 0 && wpointer[list[j]] > wpointer[v]) { list[j + 1] = list[j]; --j; }
            list[j + 1] = v;
        }
    } else {
        qsort_r(list, (size_t)len, sizeof(orc_int), cmp_slot_by_index, (void *)wpointer);
    }
}

/* threshold_and_drop, dropping.hpp:8-34 */
orc_int orc_threshold_and_drop(const double *wdata, const orc_int *wpointer, orc_int wnnz,
                               orc_int *list, orc_int n, double tau, orc_int from, orc_int to)
{
    orc_int len = 0, x;
    double z = 0.0, norm;
    if (n <= 0) return 0;

    /* vector_sparse_dynamic::norm2(begin,end), sparse_implementation.h:1087-1093: insertion order */
    for (x = 0; x < wnnz; ++x)
        if (from <= wpointer[x] && wpointer[x] < to)
            z += wdata[x] * wdata[x];
    norm = sqrt(z);

    for (x = 0; x < wnnz; ++x) {
        const orc_int i = wpointer[x];
        if (from <= i && i < to && fabs(wdata[x]) > norm * tau)
            list[len++] = x;
    }
    if (len > n) {
        orc_sort_slots_by_abs_desc(list, len, wdata);
        len = n;
    }
    sort_slots_by_index(list, len, wpointer);
    return len;
}

/* ------------------------------------------------------------------------------------------ */
/* working row: vector_sparse_dynamic / vector_sparse_ordered (sparse.h:169-323,               */
/* sparse_implementation.h:950-1093)                                                           */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
    orc_int size, nnz;
    double *data;        /* slot -> value */
    orc_int *occupancy;  /* index -> slot or -1 */
    orc_int *pointer;    /* slot -> index (insertion order) */
    orc_int *heap;       /* ordered variant: binary min-heap of slots keyed by pointer[slot] */
    orc_int heap_len;
    int ordered;
} wvec;

static void wv_init(wvec *w, orc_int m, int ordered)
{
    orc_int i;
    w->size = m; w->nnz = 0; w->ordered = ordered; w->heap_len = 0;
    /* a slot list can outgrow m only through dead slots (zero_set then re-insert); the reference
     * would write out of bounds there, we give headroom instead */
    w->data = (double *)malloc(sizeof(double) * (size_t)(2 * (size_t)m + 16));
    w->pointer = (orc_int *)malloc(sizeof(orc_int) * (size_t)(2 * (size_t)m + 16));
    w->occupancy = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m > 0 ? m : 1));
    w->heap = ordered ? (orc_int *)malloc(sizeof(orc_int) * (size_t)(2 * (size_t)m + 16)) : NULL;
    for (i = 0; i < m; ++i) w->occupancy[i] = -1;
}

static void wv_free(wvec *w)
{
    free(w->data); free(w->pointer); free(w->occupancy); free(w->heap);
}

static void wv_heap_push(wvec *w, orc_int slot)
{
    /* std::push_heap with comparator "pointer[x] > pointer[y]" => min-heap on index.  Pop order
     * among equal indices is irrelevant: the only duplicates are dead slots holding 0, which the
     * ILUT loop skips (ILUT.hpp:239-240). */
    orc_int hole = w->heap_len++;
    while (hole > 0) {
        orc_int parent = (hole - 1) / 2;
        if (w->pointer[w->heap[parent]] > w->pointer[slot]) {
            w->heap[hole] = w->heap[parent];
            hole = parent;
        } else break;
    }
    w->heap[hole] = slot;
}

static orc_int wv_pop_next_index(wvec *w)   /* sparse.h:313-322 */
{
    orc_int top, last, hole, len;
    if (w->heap_len == 0) return -1;
    top = w->heap[0];
    last = w->heap[--w->heap_len];
    len = w->heap_len;
    hole = 0;
    for (;;) {
        orc_int child = 2 * hole + 1;
        if (child >= len) break;
        if (child + 1 < len && w->pointer[w->heap[child + 1]] < w->pointer[w->heap[child]]) child++;
        if (w->pointer[w->heap[child]] < w->pointer[last]) {
            w->heap[hole] = w->heap[child];
            hole = child;
        } else break;
    }
    if (len > 0) w->heap[hole] = last;
    return top;
}

/* operator[] (insert-on-miss), sparse_implementation.h:980-994 / sparse.h:298-311; returns slot */
static orc_int wv_slot(wvec *w, orc_int j)
{
    if (w->occupancy[j] < 0) {
        w->occupancy[j] = w->nnz;
        w->pointer[w->nnz] = j;
        w->data[w->nnz] = 0.0;
        w->nnz++;
        if (w->ordered) wv_heap_push(w, w->nnz - 1);
    }
    return w->occupancy[j];
}

static void wv_zero_set_index(wvec *w, orc_int j)   /* sparse_implementation.h:1010-1021 */
{
    if (w->occupancy[j] >= 0) {
        w->data[w->occupancy[j]] = 0.0;
        w->occupancy[j] = -1;
    }
}

static void wv_zero_reset(wvec *w)   /* sparse_implementation.h:1036-1040 */
{
    orc_int i;
    for (i = 0; i < w->nnz; ++i) w->occupancy[w->pointer[i]] = -1;
    w->nnz = 0;
    w->heap_len = 0;
}

/* ------------------------------------------------------------------------------------------ */
/* ILUT (heap variant)                                                                         */
/* ------------------------------------------------------------------------------------------ */

static void mat_reserve(orc_mat *M, orc_int need, orc_int *cap)
{
    if (need > *cap) {
        orc_int nc = *cap < 1024 ? 1024 : *cap;
        while (nc < need) nc = nc + nc / 2 + 16;
        M->idx = (orc_int *)realloc(M->idx, sizeof(orc_int) * (size_t)nc);
        M->val = (double *)realloc(M->val, sizeof(double) * (size_t)nc);
        *cap = nc;
    }
}

int orc_ilut(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_int max_fill_in, double threshold, orc_mat *L, orc_mat *U, orc_int *err_row)
{
    /* ILUT_heap, ILUT.hpp:199-278.  The reference pre-reserves p(p+1)/2+(n-p)p entries in int32
     * arithmetic (:214, overflows for large n*p); the restatement grows its arrays instead and so
     * never raises "insufficient memory reserved" -- documented deviation, SURVEY appendix item 12. */
    wvec w;
    orc_int *list_L, *list_U;
    orc_int i, k, x, capL = 0, capU = 0;
    int rc = ORC_OK;

    if (max_fill_in < 1) max_fill_in = 1;
    if (max_fill_in > n) max_fill_in = n;

    wv_init(&w, n, 1);
    list_L = (orc_int *)malloc(sizeof(orc_int) * (size_t)(2 * (size_t)n + 16));
    list_U = (orc_int *)malloc(sizeof(orc_int) * (size_t)(2 * (size_t)n + 16));
    mat_init(L, n, 1, 1);
    mat_init(U, n, 1, 1);
    capL = capU = 1;

    for (i = 0; i < n; ++i) {
        double norm_wL = 0.0;
        orc_int nL, nU, kk, j, s;

        /* (2.) scatter the row, norm of the strictly-lower part in CSR order (:222-231) */
        for (k = ptr[i]; k < ptr[i + 1]; ++k) {
            s = wv_slot(&w, idx[k]);
            w.data[s] = val[k];
            if (idx[k] < i) norm_wL += val[k] * val[k];
        }
        norm_wL = sqrt(norm_wL);

        /* (3.-9.) eliminate in ascending column order (:234-255) */
        for (x = wv_pop_next_index(&w); x >= 0; x = wv_pop_next_index(&w)) {
            double wk;
            k = w.pointer[x];
            if (k >= i) break;
            if (w.data[x] == 0.0) continue;
            if (fabs(w.data[x]) < threshold * norm_wL) {
                wv_zero_set_index(&w, k);
            } else {
                w.data[x] /= U->val[U->ptr[k]];
                wk = w.data[x];
                for (j = U->ptr[k] + 1; j < U->ptr[k + 1]; j++) {
                    s = wv_slot(&w, U->idx[j]);
                    w.data[s] -= wk * U->val[j];
                }
            }
        }

        /* (10.) dropping (:259,261) */
        nL = orc_threshold_and_drop(w.data, w.pointer, w.nnz, list_L, max_fill_in - 1, threshold, 0, i);
        nU = orc_threshold_and_drop(w.data, w.pointer, w.nnz, list_U, max_fill_in - 1, threshold, i + 1, n);

        /* (11.) L row = kept entries then (i, 1.0)  (append_row_with_suffix :3212-3230) */
        kk = L->ptr[i];
        mat_reserve(L, kk + nL + 1, &capL);
        for (j = 0; j < nL; ++j) { L->val[kk] = w.data[list_L[j]]; L->idx[kk++] = w.pointer[list_L[j]]; }
        L->val[kk] = 1.0; L->idx[kk++] = i;
        L->ptr[i + 1] = kk;

        /* (12.) U row = (i, w[i]) then kept entries (append_row_with_prefix :3190-3210);
         * w[i] inserts a zero slot when the row has no diagonal */
        s = wv_slot(&w, i);
        kk = U->ptr[i];
        mat_reserve(U, kk + nU + 1, &capU);
        U->val[kk] = w.data[s]; U->idx[kk++] = i;
        for (j = 0; j < nU; ++j) { U->val[kk] = w.data[list_U[j]]; U->idx[kk++] = w.pointer[list_U[j]]; }
        U->ptr[i + 1] = kk;

        if (U->val[U->ptr[i]] == 0.0) {      /* :269-270 */
            if (err_row) *err_row = i;
            rc = ORC_ERR_ZERO_PIVOT;
            break;
        }
        wv_zero_reset(&w);
    }
    wv_free(&w); free(list_L); free(list_U);
    if (rc != ORC_OK) { orc_free_mat(L); orc_free_mat(U); return rc; }

    L->nnz = L->ptr[n]; U->nnz = U->ptr[n];
    mat_compress(L, 0.0);   /* :275-276 */
    mat_compress(U, 0.0);

    if (!is_csr) {   /* binding.cpp:440-444 / preconditioner_implementation.h:999-1001 */
        mat_swap(L, U);
        L->is_csr = 0;
        U->is_csr = 0;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* IChol(0)                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* natural_triangular_part(increasing), sparse_implementation.h:2568-2592 (orientation-agnostic) */
static void natural_triangular_part(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val,
                                    int increasing, orc_mat *T, int is_csr)
{
    orc_int i, k, cur = 0;
    mat_init(T, n, ptr[n], is_csr);
    for (i = 0; i < n; ++i) {
        for (k = ptr[i]; k < ptr[i + 1]; ++k) {
            const orc_int j = idx[k];
            if ((increasing && j <= i) || (!increasing && j >= i)) {
                T->idx[cur] = j;
                T->val[cur++] = val[k];
            }
        }
        T->ptr[i + 1] = cur;
    }
    T->nnz = cur;
}

/* sparse_dot_product, IChol.hpp:16-28 */
static double sparse_dot_product(orc_int l1, orc_int u1, orc_int l2, orc_int u2, const orc_int *indices, const double *data)
{
    double result = 0.0;
    while (l1 < u1 && l2 < u2) {
        if (indices[l1] == indices[l2]) {
            result += data[l1] * data[l2];
            l1++; l2++;
        } else if (indices[l1] < indices[l2])
            l1++;
        else
            l2++;
    }
    return result;
}

int orc_ichol0(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
               orc_mat *L)
{
    /* IChol0, IChol.hpp:63-73: lower part in major order (j <= i), relabelled ROW */
    orc_int i, k;
    double *nd;
    (void)is_csr;
    natural_triangular_part(n, ptr, idx, val, 1, L, 1);
    nd = (double *)malloc(sizeof(double) * (size_t)(L->nnz > 0 ? L->nnz : 1));

    /* compute_ichol0, IChol.hpp:33-59 */
    for (i = 0; i < n; ++i) {
        for (k = L->ptr[i]; k < L->ptr[i + 1]; ++k) {
            const orc_int j = L->idx[k];
            const double dp = sparse_dot_product(L->ptr[i], L->ptr[i + 1] - 1, L->ptr[j], L->ptr[j + 1] - 1, L->idx, nd);
            const double A_ij = L->val[k];
            if (j < i) {
                const double L_jj = nd[L->ptr[j + 1] - 1];
                nd[k] = (A_ij - dp) / L_jj;
            } else if (j == i)
                nd[k] = sqrt(A_ij - dp);
            else { free(nd); orc_free_mat(L); return ORC_ERR_NOT_TRIANGULAR; }
        }
    }
    memcpy(L->val, nd, sizeof(double) * (size_t)L->nnz);
    free(nd);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ICholT                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* update_triangular_fields, ILUC.hpp:37-63 */
static void update_triangular_fields(orc_int k, const orc_int *pointer, const orc_int *indices, orc_int *list, orc_int *first)
{
    orc_int h, i, j;
    for (h = list[k]; h != -1; h = list[h]) first[h] += 1;
    first[k] = pointer[k] + 1;
    h = list[k];
    if (pointer[k] + 1 < pointer[k + 1]) {
        j = indices[pointer[k] + 1];
        list[k] = list[j];
        list[j] = k;
    }
    while (h != -1) {
        i = h;
        h = list[i];
        if (first[i] < pointer[i + 1]) {
            j = indices[first[i]];
            list[i] = list[j];
            list[j] = i;
        }
    }
}

int orc_icholt(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
               orc_int add_fill_in, double threshold, orc_mat *L)
{
    /* ICholT, IChol.hpp:158-164: keep j >= i in major order, relabel as CSC lower triangle */
    orc_mat A;
    const orc_int m = n;
    orc_int reserved, j, x, k;
    orc_int *firstL, *listL, *listw;
    double *D;
    wvec w;
    int rc = ORC_OK;
    (void)is_csr;

    natural_triangular_part(n, ptr, idx, val, 0, &A, 0);

    /* ICholT_tri, IChol.hpp:78-155 */
    {
        long a = (long)A.nnz + (long)(add_fill_in > 0 ? add_fill_in : 0) * (long)m;   /* :85-87 (int32 in the reference) */
        long b = (long)((double)A.nnz * 10.0);
        long r = a < b ? a : b;
        reserved = (orc_int)r;
    }
    firstL = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m > 0 ? m : 1));
    listL = (orc_int *)malloc(sizeof(orc_int) * (size_t)(m > 0 ? m : 1));
    listw = (orc_int *)malloc(sizeof(orc_int) * (size_t)(2 * (size_t)m + 16));
    D = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    for (j = 0; j < m; ++j) { listL[j] = -1; firstL[j] = 0; }   /* std::vector<Integer>(m) zero-inits firstL */
    mat_init(L, m, reserved, 0);
    wv_init(&w, m, 0);

    for (j = 0; j < m; ++j) {
        double L_jj;
        orc_int s, col_len, nkeep, kk;

        if (A.ptr[j] >= A.ptr[j + 1] || A.idx[A.ptr[j]] != j) { rc = ORC_ERR_NOT_TRIANGULAR; break; }   /* :105-107 */

        wv_zero_reset(&w);
        for (x = A.ptr[j]; x < A.ptr[j + 1]; ++x) {
            s = wv_slot(&w, A.idx[x]);
            w.data[s] = A.val[x];
        }
        D[j] += A.val[A.ptr[j]];
        L_jj = sqrt(D[j]);
        w.data[wv_slot(&w, j)] = L_jj;

        for (k = listL[j]; k != -1; k = listL[k]) {   /* :120-132, linked-list order */
            const double L_jk = L->val[firstL[k]];
            x = firstL[k];
            if (L->idx[x] == j) ++x;
            for (; x < L->ptr[k + 1]; ++x) {
                s = wv_slot(&w, L->idx[x]);
                w.data[s] -= L->val[x] * L_jk;
            }
        }

        for (x = 0; x < w.nnz; ++x) {   /* :135-141 */
            const orc_int i = w.pointer[x];
            if (i > j) {
                w.data[x] /= L_jj;
                D[i] -= w.data[x] * w.data[x];
            }
        }

        col_len = A.ptr[j + 1] - A.ptr[j];
        nkeep = orc_threshold_and_drop(w.data, w.pointer, w.nnz, listw, col_len + add_fill_in, threshold, j, m);

        /* append_row, sparse_implementation.h:3170-3186 */
        kk = L->ptr[j];
        if (kk + nkeep > reserved) { rc = ORC_ERR_MEMORY; break; }
        for (x = 0; x < nkeep; ++x) { L->val[kk] = w.data[listw[x]]; L->idx[kk++] = w.pointer[listw[x]]; }
        L->ptr[j + 1] = kk;

        update_triangular_fields(j, L->ptr, L->idx, listL, firstL);
    }

    wv_free(&w); free(firstL); free(listL); free(listw); free(D);
    orc_free_mat(&A);
    if (rc != ORC_OK) { orc_free_mat(L); return rc; }
    L->nnz = L->ptr[m];
    mat_compress(L, -1.0);   /* :153 */
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ILUC (Crout ILU, Li/Saad/Chow)                                                              */
/* ------------------------------------------------------------------------------------------ */

/* initialize_sparse_matrix_fields, ILUC.hpp:74-84: list[i] chains the rows whose first unprocessed entry lies in
 * the same column; head[j] is the first row of column j's chain */
static void initialize_sparse_matrix_fields(orc_int n, const orc_int *pointer, const orc_int *indices, orc_int *list, orc_int *head,
                                            orc_int *first)
{
    orc_int k;
    for (k = 0; k < n; ++k) head[k] = -1;
    for (k = 0; k < n; ++k) {
        first[k] = pointer[k];
        if (pointer[k] < pointer[k + 1]) { list[k] = head[indices[pointer[k]]]; head[indices[pointer[k]]] = k; }
    }
}

/* update_sparse_matrix_fields, ILUC.hpp:86-101 */
static void update_sparse_matrix_fields(orc_int k, const orc_int *pointer, const orc_int *indices, orc_int *list, orc_int *head,
                                        orc_int *first)
{
    orc_int h, i;
    for (h = head[k]; h != -1; h = list[h]) first[h] += 1;
    h = head[k];
    while (h != -1) {
        i = h;
        h = list[i];
        if (first[i] < pointer[i + 1]) { list[i] = head[indices[first[i]]]; head[indices[first[i]]] = i; }
    }
}

/* append_row_with_prefix, sparse_implementation.h:3189-3208 */
static int append_with_prefix(orc_mat *M, orc_int reserved, orc_int i, const wvec *w, const orc_int *list, orc_int len, double prefix)
{
    orc_int kk = M->ptr[i], x;
    if ((long)kk + (long)len + 1 > (long)reserved) return ORC_ERR_MEMORY;
    M->val[kk] = prefix; M->idx[kk++] = i;
    for (x = 0; x < len; ++x) { M->val[kk] = w->data[list[x]]; M->idx[kk++] = w->pointer[list[x]]; }
    M->ptr[i + 1] = kk;
    return ORC_OK;
}

int orc_iluc(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
             orc_int max_fill_in, double threshold, orc_mat *Lout, orc_mat *Uout, orc_int *err_row)
{
    /* ILUC2, ILUC.hpp:112-207, on the major-order view of the arrays (for a COLUMN matrix: of A^T); orientation of the results
     * and their interchange for COLUMN input: binding.cpp:449-460 */
    const orc_int m = n;
    orc_int k, j, h, x, reserved;
    orc_int *firstU, *listU, *firstL, *listL, *listA, *headA, *firstA, *listw, *listz;
    wvec z, w;
    orc_mat L, U;
    int rc = ORC_OK;
    long nnzA = ptr[n];
    {
        /* :120 (Integer arithmetic in the reference; the product is taken in 64 bits here) */
        long a = (long)max_fill_in * (long)n, b = (long)(10.0 * (double)nnzA);
        long r = a < b ? a : b;
        reserved = (orc_int)(r > 0 ? r : 0);
    }
    if (err_row) *err_row = -1;
    firstU = (orc_int *)calloc((size_t)m + 1, sizeof(orc_int)); listU = (orc_int *)malloc(sizeof(orc_int) * ((size_t)m + 1));
    firstL = (orc_int *)calloc((size_t)m + 1, sizeof(orc_int)); listL = (orc_int *)malloc(sizeof(orc_int) * ((size_t)m + 1));
    listA = (orc_int *)calloc((size_t)m + 1, sizeof(orc_int)); headA = (orc_int *)malloc(sizeof(orc_int) * ((size_t)m + 1));
    firstA = (orc_int *)calloc((size_t)m + 1, sizeof(orc_int));
    listw = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)m + 16)); listz = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)m + 16));
    mat_init(&L, m, reserved > 0 ? reserved : 1, !is_csr);       /* other_orientation(A.orientation), :129 */
    mat_init(&U, m, reserved > 0 ? reserved : 1, is_csr);
    wv_init(&z, n, 0); wv_init(&w, m, 0);
    if (max_fill_in < 1) max_fill_in = 1;                        /* :133 (after the reservation) */
    initialize_sparse_matrix_fields(m, ptr, idx, listA, headA, firstA);
    for (k = 0; k < m; ++k) { listL[k] = -1; listU[k] = -1; }

    for (k = 0; k < m; ++k) {
        orc_int nw, nz;
        double U_kk;
        wv_zero_reset(&z);                                        /* (2.) */
        for (j = firstA[k]; j < ptr[k + 1]; ++j) z.data[wv_slot(&z, idx[j])] = val[j];
        for (h = listL[k]; h != -1; h = listL[h]) {               /* (3.)-(5.) */
            const double L_kh = L.val[firstL[h]];
            for (j = firstU[h]; j < U.ptr[h + 1]; ++j) { x = wv_slot(&z, U.idx[j]); z.data[x] -= L_kh * U.val[j]; }
        }
        wv_zero_reset(&w);                                        /* (6.) */
        for (h = headA[k]; h != -1; h = listA[h])
            if (h > k) w.data[wv_slot(&w, h)] = val[firstA[h]];
        for (h = listU[k]; h != -1; h = listU[h]) {               /* (7.)-(9.) */
            const double U_hk = U.val[firstU[h]];
            for (j = firstL[h]; j < L.ptr[h + 1]; ++j) { x = wv_slot(&w, L.idx[j]); w.data[x] -= U_hk * L.val[j]; }
        }
        if (z.occupancy[k] < 0) { rc = ORC_ERR_ZERO_PIVOT; if (err_row) *err_row = k; break; }    /* :174-175 */
        nw = orc_threshold_and_drop(w.data, w.pointer, w.nnz, listw, max_fill_in - 1, threshold, k + 1, m);   /* (10.) */
        nz = orc_threshold_and_drop(z.data, z.pointer, z.nnz, listz, max_fill_in - 1, threshold, k + 1, n);   /* (11.) */
        rc = append_with_prefix(&U, reserved, k, &z, listz, nz, z.data[z.occupancy[k]]);                         /* (12.) */
        if (rc) break;
        U_kk = U.val[U.ptr[k]];
        for (x = 0; x < nw; ++x) w.data[listw[x]] /= U_kk;                                                       /* (13.) */
        rc = append_with_prefix(&L, reserved, k, &w, listw, nw, 1.0);
        if (rc) break;
        update_sparse_matrix_fields(k, ptr, idx, listA, headA, firstA);
        update_triangular_fields(k, L.ptr, L.idx, listL, firstL);
        update_triangular_fields(k, U.ptr, U.idx, listU, firstU);
    }
    wv_free(&z); wv_free(&w);
    free(firstU); free(listU); free(firstL); free(listL); free(listA); free(headA); free(firstA); free(listw); free(listz);
    if (rc != ORC_OK) { orc_free_mat(&L); orc_free_mat(&U); return rc; }
    L.nnz = L.ptr[m]; U.nnz = U.ptr[m];
    mat_compress(&L, -1.0); mat_compress(&U, -1.0);               /* :205-206 */
    if (is_csr) { *Lout = L; *Uout = U; }
    else { *Lout = U; *Uout = L; }                                /* binding.cpp:456-457 */
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* triangular solves + apply                                                                   */
/* ------------------------------------------------------------------------------------------ */

void orc_trisolve(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr,
                  int form, int use, double *x)
{
    /* matrix_sparse::triangular_solve, sparse_implementation.h:4040-4087 */
    orc_int j, k;
    const int row = is_csr != 0;
    if ((form == ORC_LOWER && row && use == ORC_ID) || (form == ORC_UPPER && !row && use == ORC_TRANSPOSE)) {
        for (k = 0; k < n; k++) {                                   /* T1: forward gather, diagonal last */
            for (j = ptr[k]; j < ptr[k + 1] - 1; j++) x[k] -= val[j] * x[idx[j]];
            x[k] /= val[ptr[k + 1] - 1];
        }
        return;
    }
    if ((form == ORC_LOWER && !row && use == ORC_ID) || (form == ORC_UPPER && row && use == ORC_TRANSPOSE)) {
        for (k = 0; k < n; k++) {                                   /* T2: forward scatter, diagonal first */
            x[k] /= val[ptr[k]];
            for (j = ptr[k] + 1; j < ptr[k + 1]; j++) x[idx[j]] -= val[j] * x[k];
        }
        return;
    }
    if ((form == ORC_UPPER && row && use == ORC_ID) || (form == ORC_LOWER && !row && use == ORC_TRANSPOSE)) {
        for (k = n - 1; k >= 0; k--) {                              /* T3: backward gather, diagonal first */
            for (j = ptr[k] + 1; j < ptr[k + 1]; j++) x[k] -= val[j] * x[idx[j]];
            x[k] /= val[ptr[k]];
        }
        return;
    }
    for (k = n - 1; k >= 0; k--) {                                  /* T4: backward scatter, diagonal last */
        x[k] /= val[ptr[k + 1] - 1];
        for (j = ptr[k]; j < ptr[k + 1] - 1; j++) x[idx[j]] -= val[j] * x[k];
    }
}

void orc_apply_lu(const orc_mat *L, const orc_mat *U, int use, double *x)
{
    /* split_preconditioner::apply_preconditioner_only, preconditioner_implementation.h:103-111 */
    if (use == ORC_ID) {
        orc_trisolve(L->n, L->ptr, L->idx, L->val, L->is_csr, ORC_LOWER, ORC_ID, x);
        orc_trisolve(U->n, U->ptr, U->idx, U->val, U->is_csr, ORC_UPPER, ORC_ID, x);
    } else {
        orc_trisolve(U->n, U->ptr, U->idx, U->val, U->is_csr, ORC_UPPER, ORC_TRANSPOSE, x);
        orc_trisolve(L->n, L->ptr, L->idx, L->val, L->is_csr, ORC_LOWER, ORC_TRANSPOSE, x);
    }
}

void orc_apply_llt(const orc_mat *L, int use, double *x)
{
    /* :381-394: left = solve(LOWER, use), right = solve(LOWER, other_usage(use)) */
    if (use == ORC_ID) {
        orc_trisolve(L->n, L->ptr, L->idx, L->val, L->is_csr, ORC_LOWER, ORC_ID, x);
        orc_trisolve(L->n, L->ptr, L->idx, L->val, L->is_csr, ORC_LOWER, ORC_TRANSPOSE, x);
    } else {
        /* TRANSPOSE: right first with other_usage(TRANSPOSE)=ID, then left with TRANSPOSE */
        orc_trisolve(L->n, L->ptr, L->idx, L->val, L->is_csr, ORC_LOWER, ORC_ID, x);
        orc_trisolve(L->n, L->ptr, L->idx, L->val, L->is_csr, ORC_LOWER, ORC_TRANSPOSE, x);
    }
}

/* ========================================================================================== */
/* Multilevel ILU++ without pivoting: SURVEY section 8(f) rank 3, the precon_parameter 10     */
/* family (parameters_implementation.h:927-934: PERMUTE_ROWS 0, TOTAL_PIV 0, piv_tol 0,       */
/* SMALL_PIVOT_TERMINATES) = multilevelILUCDPPreconditioner::make_preprocessed_multilevelILUCDP */
/* (preconditioner_implementation.h:1350-1665) over matrix_sparse::partialILUC                 */
/* (ILUCDP.hpp:1405-2231) with the preprocessing of matrix_sparse::preprocess                  */
/* (sparse_implementation.h:5214-5460) and the apply of :433-488.                              */
/* Knobs outside that family (pivoting, inverse / weighted / positional dropping, improved     */
/* Schur complement, bounded fill) are NOT restated: orc_ml_create refuses them.               */
/* ========================================================================================== */

struct orc_ml_level {
    orc_int n;
    orc_mat L, U;                 /* Precond_left (COLUMN, 1 first), Precond_right (ROW, 1 first) */
    double *D;                    /* Precond_middle: 1 / pivot */
    orc_int *perm_rows, *perm_cols, *inv_perm_rows, *inv_perm_cols;
    double *D_l, *D_r;
    orc_int zero_pivots;
};

struct orc_ml {
    orc_int n;
    int nlevels;
    struct orc_ml_level *lev;
};

static void mat_copy(orc_mat *dst, const orc_mat *src)
{
    mat_init(dst, src->n, src->nnz, src->is_csr);
    memcpy(dst->ptr, src->ptr, sizeof(orc_int) * ((size_t)src->n + 1));
    memcpy(dst->idx, src->idx, sizeof(orc_int) * (size_t)src->nnz);
    memcpy(dst->val, src->val, sizeof(double) * (size_t)src->nnz);
    dst->nnz = src->nnz;
}

/* matrix_sparse::change_orientation, sparse_implementation.h:2543-2566: same matrix, other storage order */
static void mat_change_orientation(const orc_mat *A, orc_mat *B)
{
    orc_int i, j, *counter;
    const orc_int n = A->n, nnz = A->ptr[n];
    mat_init(B, n, nnz, !A->is_csr);
    for (i = 0; i < nnz; ++i) B->ptr[1 + A->idx[i]]++;
    for (i = 1; i <= n; ++i) B->ptr[i] += B->ptr[i - 1];
    counter = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));
    for (i = 0; i < n; ++i)
        for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) {
            const orc_int l = A->idx[j], k = B->ptr[l] + counter[l];
            B->val[k] = A->val[j]; B->idx[k] = i; counter[l]++;
        }
    free(counter);
    B->nnz = nnz;
}

/* vector_dense<T>::quicksort(index_list&, left, right), sparse_implementation.h:471-505: the reference's own (unstable) quicksort
 * on the values, the list follows -- the order of equal values is part of the PQ permutation */
static void vec_quicksort(double *data, orc_int *list, orc_int left, orc_int right)
{
    orc_int i, j;
    double m;
    if (left < right) {
        m = data[left]; i = left; j = right;
        while (i <= j) {
            while (data[i] < m) i++;
            while (data[j] > m) j--;
            if (i <= j) {
                const double t = data[i]; const orc_int u = list[i];
                data[i] = data[j]; data[j] = t; list[i] = list[j]; list[j] = u;
                i++; j--;
            }
        }
        vec_quicksort(data, list, left, j);
        vec_quicksort(data, list, i, right);
    }
}

/* matrix_sparse::normal_order, :3381-3405: every row by increasing index (the indices of a row are distinct) */
static int cmp_idx_pair(const void *a, const void *b)
{
    const orc_int x = *(const orc_int *)a, y = *(const orc_int *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static void mat_normal_order(orc_mat *M)
{
    orc_int k, j, maxlen = 0;
    struct pr { orc_int idx; orc_int pos; } *buf;
    double *tmp;
    for (k = 0; k < M->n; ++k) if (M->ptr[k + 1] - M->ptr[k] > maxlen) maxlen = M->ptr[k + 1] - M->ptr[k];
    buf = (struct pr *)malloc(sizeof(struct pr) * (size_t)(maxlen > 0 ? maxlen : 1));
    tmp = (double *)malloc(sizeof(double) * (size_t)(maxlen > 0 ? maxlen : 1));
    for (k = 0; k < M->n; ++k) {
        const orc_int b = M->ptr[k], len = M->ptr[k + 1] - b;
        for (j = 0; j < len; ++j) { buf[j].idx = M->idx[b + j]; buf[j].pos = j; tmp[j] = M->val[b + j]; }
        qsort(buf, (size_t)len, sizeof(struct pr), cmp_idx_pair);
        for (j = 0; j < len; ++j) { M->idx[b + j] = buf[j].idx; M->val[b + j] = tmp[buf[j].pos]; }
    }
    free(buf); free(tmp);
}

/* matrix_sparse::permute(p1,p2,ip1,ip2) for a ROW matrix, :5570-5573 -> permute_along_with_perm_and_against_orientation_with_invperm
 * :5533-5548: new row i = old row perm_along[i], every column index c becomes invperm_against[c], rows re-sorted */
static void mat_permute_rows_cols(orc_mat *A, const orc_int *perm_along, const orc_int *invperm_against)
{
    orc_mat H;
    orc_int i, j, counter = 0;
    mat_init(&H, A->n, A->nnz, A->is_csr);
    for (i = 0; i < A->n; ++i) {
        H.ptr[i] = counter;
        for (j = A->ptr[perm_along[i]]; j < A->ptr[perm_along[i] + 1]; ++j) {
            H.val[counter] = A->val[j];
            H.idx[counter] = invperm_against[A->idx[j]];
            counter++;
        }
    }
    H.ptr[A->n] = counter;
    H.nnz = counter;
    mat_normal_order(&H);
    orc_free_mat(A);
    *A = H;
}

static void perm_invert(orc_int *inv, const orc_int *perm, orc_int n)      /* index_list::invert, :6140-6144 */
{
    orc_int i;
    for (i = 0; i < n; ++i) inv[perm[i]] = i;
}
static void perm_compose_right(orc_int *P, const orc_int *Q, orc_int n)    /* P := P o Q: P[i] = P[Q[i]], :6165-6183 */
{
    orc_int i, *H = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
    for (i = 0; i < n; ++i) H[i] = P[Q[i]];
    memcpy(P, H, sizeof(orc_int) * (size_t)n);
    free(H);
}
static void vec_permute(double *x, const orc_int *perm, orc_int n)          /* vector_dense::permute(perm), :590-597 */
{
    orc_int i;
    double *H = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (i = 0; i < n; ++i) H[i] = x[perm[i]];
    memcpy(x, H, sizeof(double) * (size_t)n);
    free(H);
}

/* matrix_sparse::ddPQ(P, Q, tau), :4571-4635 (Saad's greedy PQ ordering); P, Q are the INVERSE permutations */
static orc_int mat_ddPQ(const orc_mat *A, orc_int *P, orc_int *Q, double tau)
{
    const orc_int n = A->n;
    orc_int j, k, count, Qcount, pos;
    orc_int *I = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
    orc_int *J = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
    orc_int *J2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
    double *W = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (k = 0; k < n; ++k) { P[k] = -1; Q[k] = -1; I[k] = k; }
    for (k = 0; k < n; ++k) {
        double current_max = 0.0, divisor;
        W[k] = 0.0; J[k] = 0;
        for (j = A->ptr[k]; j < A->ptr[k + 1]; ++j) {
            W[k] += fabs(A->val[j]);
            if (fabs(A->val[j]) > current_max) { current_max = fabs(A->val[j]); J[k] = A->idx[j]; }
        }
        divisor = W[k] * (double)(A->ptr[k + 1] - A->ptr[k]);
        if (divisor == 0.0) W[k] = 0.0;
        else W[k] = -current_max / divisor;
    }
    vec_quicksort(W, I, 0, n - 1);
    for (k = 0; k < n; ++k) J2[k] = J[I[k]];                                   /* permute_vec(J, I) */
    count = -1;
    for (k = 0; k < n; ++k)
        if (P[I[k]] == -1 && Q[J2[k]] == -1 && -W[k] >= tau) { count++; P[I[k]] = count; Q[J2[k]] = count; }
    pos = Qcount = count;
    for (k = 0; k < n; ++k) if (P[k] < 0) { count++; P[k] = count; }
    for (k = 0; k < n; ++k) if (Q[k] < 0) { Qcount++; Q[k] = Qcount; }
    free(I); free(J); free(J2); free(W);
    return pos + 1;
}

/* ---- maximum-weight perfect matching with scalings: find_pmwm, pmwm_implementation.h:385-537 (class sapTree :37-383) ---- */
typedef struct { orc_int index; double value, weight; } pm_dist;
/* std::priority_queue<dist, vector<dist>, greater<dist>> = libstdc++'s push_heap / pop_heap (bits/stl_heap.h) with comp(a, b) = a.value > b.value */
static void pm_push_heap(pm_dist *first, long hole, long top, pm_dist v)
{
    long parent = (hole - 1) / 2;
    while (hole > top && first[parent].value > v.value) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = v;
}
static void pm_adjust_heap(pm_dist *first, long hole, long len, pm_dist v)
{
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (first[child].value > first[child - 1].value) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    pm_push_heap(first, hole, top, v);
}
typedef struct { pm_dist *a; long len, cap; } pm_queue;
static void pmq_push(pm_queue *q, pm_dist v)
{
    if (q->len == q->cap) { q->cap = q->cap ? 2 * q->cap : 64; q->a = (pm_dist *)realloc(q->a, sizeof(pm_dist) * (size_t)q->cap); }
    q->len++;
    pm_push_heap(q->a, q->len - 1, 0, v);
}
static void pmq_pop(pm_queue *q)
{
    if (q->len > 1) {
        const pm_dist v = q->a[q->len - 1];
        q->a[q->len - 1] = q->a[0];
        pm_adjust_heap(q->a, 0, q->len - 1, v);
    }
    q->len--;
}

static int mat_pmwm(const orc_mat *A, orc_int *mate_row, orc_int *mate_col, double *inv_row, double *inv_col)
{
    const orc_int dim = A->n, nz = A->ptr[A->n];
    const orc_int *ptr = A->ptr, *idx = A->idx;
    orc_int i, row, col, r, s;
    int ok = 1;
    double *u = (double *)calloc((size_t)dim + 1, sizeof(double)), *v = (double *)calloc((size_t)dim + 1, sizeof(double));
    double *comp = (double *)malloc(sizeof(double) * (size_t)(nz > 0 ? nz : 1)), *amax = (double *)calloc((size_t)dim + 1, sizeof(double));
    orc_int *row_pointer = (orc_int *)calloc((size_t)dim + 1, sizeof(orc_int));
    double *cand_weights = (double *)calloc((size_t)dim + 1, sizeof(double)), *weights = (double *)calloc((size_t)dim + 1, sizeof(double));
    wvec checked, rdist;
    pm_queue q = {NULL, 0, 0};
    wv_init(&checked, dim, 0); wv_init(&rdist, dim, 0);
    for (i = 0; i < dim; ++i) { mate_row[i] = -1; mate_col[i] = -1; inv_row[i] = 0.0; inv_col[i] = 0.0; }
    /* transform_and_copy_data, :362-372 */
    for (row = 0; row < dim; ++row)
        for (i = ptr[row]; i < ptr[row + 1]; ++i) if (amax[row] < fabs(A->val[i])) amax[row] = fabs(A->val[i]);
    for (row = 0; row < dim; ++row)
        for (i = ptr[row]; i < ptr[row + 1]; ++i) comp[i] = log(amax[row] / fabs(A->val[i]));
    /* dual_initialization, :171-200 */
    for (i = 0; i < dim; ++i) { v[i] = -1; u[i] = -1; }
    for (i = 0; i < nz; ++i) { col = idx[i]; if (v[col] > comp[i] || v[col] == -1) v[col] = comp[i]; }
    for (row = 0; row < dim; ++row)
        for (i = ptr[row]; i < ptr[row + 1]; ++i)
            if (u[row] > comp[i] - v[idx[i]] || u[row] == -1) u[row] = comp[i] - v[idx[i]];
    /* matching_initialization, :203-250 */
    for (row = 0; row < dim; ++row)
        for (i = ptr[row]; i < ptr[row + 1]; ++i) {
            col = idx[i];
            if (mate_col[col] == -1 && (comp[i] - u[row] - v[col] == 0)) { mate_row[row] = col; mate_col[col] = row; weights[col] = comp[i]; break; }
        }
    for (row = 0; row < dim; ++row) {
        if (mate_row[row] != -1) continue;
        for (i = ptr[row]; i < ptr[row + 1]; ++i) {
            col = idx[i];
            if (mate_col[col] != -1 && (comp[i] - u[row] - v[col] == 0)) {
                const orc_int row_new = mate_col[col];
                orc_int j;
                for (j = ptr[row_new]; j < ptr[row_new + 1]; ++j) {
                    const orc_int col_new = idx[j];
                    if (mate_col[col_new] == -1 && (comp[j] - u[row_new] - v[col_new] == 0)) {
                        mate_row[row] = col; mate_col[col] = row;
                        mate_row[row_new] = col_new; mate_col[col_new] = row_new;
                        weights[col_new] = comp[j]; weights[col] = comp[i];
                        break;
                    }
                }
            }
            if (mate_row[row] != -1) break;
        }
    }
    for (r = 0; r < dim && ok; ++r) {
        orc_int isap = -1, jsap = -1, j, k;
        double lsap = -1, lsp = 0;
        if (mate_row[r] != -1) continue;
        /* reset(r), :133-141 */
        q.len = 0;
        wv_zero_reset(&checked); wv_zero_reset(&rdist);
        /* find_sap, :287-355 */
        i = r;
        for (;;) {
            orc_int d;
            pm_dist md;
            orc_int j_min;
            for (d = ptr[i]; d < ptr[i + 1]; ++d) {
                j = idx[d];
                if (checked.occupancy[j] < 0) {
                    const double weight = comp[d];
                    const double dnew = lsp + weight - u[i] - v[j];
                    if (lsap == -1 || dnew < lsap) {
                        if (mate_col[j] == -1) { lsap = dnew; cand_weights[j] = weight; jsap = j; isap = i; }
                        else if (rdist.occupancy[j] < 0 || dnew < rdist.data[rdist.occupancy[j]]) {
                            pm_dist t;
                            rdist.data[wv_slot(&rdist, j)] = dnew;
                            row_pointer[mate_col[j]] = i;
                            t.index = j; t.value = dnew; t.weight = weight;
                            pmq_push(&q, t);
                        }
                    }
                }
            }
            if (q.len == 0) break;
            do { md = q.a[0]; j_min = md.index; pmq_pop(&q); } while (checked.occupancy[j_min] >= 0 && q.len != 0);
            if (q.len == 0 && checked.occupancy[j_min] >= 0) break;
            lsp = md.value;
            if (lsap != -1 && lsap <= lsp) break;
            cand_weights[j_min] = md.weight;
            checked.data[wv_slot(&checked, j_min)] = 1;
            i = mate_col[j_min];
        }
        if (lsap == -1 || jsap == -1) { ok = 0; break; }                        /* :460-471: no perfect matching */
        /* augment(mate_row, mate_col, isap, jsap), :144-168 */
        i = isap; j = jsap;
        mate_col[j] = i;
        while (i != r) {
            weights[j] = cand_weights[j];
            k = mate_row[i];
            mate_row[i] = j;
            j = k;
            i = row_pointer[i];
            mate_col[j] = i;
        }
        mate_row[r] = j;
        weights[j] = cand_weights[j];
        /* dual_update, :253-284 */
        for (s = 0; s < checked.nnz; ++s) { col = checked.pointer[s]; v[col] = v[col] + rdist.data[wv_slot(&rdist, col)] - lsap; }
        i = isap; j = jsap;
        while (i != r) {
            checked.data[wv_slot(&checked, j)] = 0;
            u[i] = weights[j] - v[j];
            i = row_pointer[i]; j = mate_row[i];
        }
        checked.data[wv_slot(&checked, j)] = 0; u[r] = weights[j] - v[j];
        for (s = 0; s < checked.nnz; ++s) { col = checked.pointer[s]; u[mate_col[col]] = weights[col] - v[col]; }
    }
    if (!ok) {
        for (s = 0; s < dim; ++s) { inv_row[s] = 1.0; inv_col[s] = 1.0; mate_row[s] = s; mate_col[s] = s; }
    } else {
        for (s = 0; s < dim; ++s) { inv_row[s] = amax[s] / exp(u[s]); inv_col[s] = exp(-v[s]); }                 /* :512-515 */
    }
    wv_free(&checked); wv_free(&rdist); free(q.a);
    free(u); free(v); free(comp); free(amax); free(row_pointer); free(cand_weights); free(weights);
    return ok;
}

/* ---- sorted_vector (arrays_implementation.h:43-155): a std::multimap<Real, Integer> -- equal keys in insertion order -- as a binary
 * heap on (key, insertion number) with entries that are left behind when an index is re-inserted or removed ---- */
typedef struct { double key; long seq; orc_int idx; } sv_ent;
typedef struct { sv_ent *a; long len, cap, counter; long *cur; double *key; char *used; orc_int n; } sorted_vec;
static int sv_less(const sv_ent *x, const sv_ent *y) { return x->key < y->key || (x->key == y->key && x->seq < y->seq); }
static void sv_push(sorted_vec *w, sv_ent e)
{
    long h;
    if (w->len == w->cap) { w->cap = w->cap ? 2 * w->cap : 64; w->a = (sv_ent *)realloc(w->a, sizeof(sv_ent) * (size_t)w->cap); }
    h = w->len++;
    while (h > 0 && sv_less(&e, &w->a[(h - 1) / 2])) { w->a[h] = w->a[(h - 1) / 2]; h = (h - 1) / 2; }
    w->a[h] = e;
}
static void sv_pop(sorted_vec *w)
{
    sv_ent e = w->a[--w->len];
    long h = 0;
    for (;;) {
        long c = 2 * h + 1;
        if (c >= w->len) break;
        if (c + 1 < w->len && sv_less(&w->a[c + 1], &w->a[c])) c++;
        if (!sv_less(&w->a[c], &e)) break;
        w->a[h] = w->a[c]; h = c;
    }
    if (w->len > 0) w->a[h] = e;
}
static void sv_insert(sorted_vec *w, orc_int pos, double val)
{
    sv_ent e;
    e.key = val; e.seq = ++w->counter; e.idx = pos;
    w->cur[pos] = e.seq; w->key[pos] = val; w->used[pos] = 1;
    sv_push(w, e);
}
static void sv_resize(sorted_vec *w, orc_int n)
{
    orc_int k;
    w->len = 0;
    for (k = 0; k < n; ++k) sv_insert(w, k, 0.0);
}
static void sv_init(sorted_vec *w, orc_int n)
{
    memset(w, 0, sizeof(*w));
    w->n = n;
    w->cur = (long *)calloc((size_t)n + 1, sizeof(long)); w->key = (double *)calloc((size_t)n + 1, sizeof(double)); w->used = (char *)calloc((size_t)n + 1, 1);
    sv_resize(w, n);
}
static void sv_free(sorted_vec *w) { free(w->a); free(w->cur); free(w->key); free(w->used); }
static orc_int sv_index_min(sorted_vec *w)
{
    while (w->len > 0 && !(w->used[w->a[0].idx] && w->cur[w->a[0].idx] == w->a[0].seq)) sv_pop(w);
    return w->len > 0 ? w->a[0].idx : -1;
}
static void sv_remove(sorted_vec *w, orc_int k) { w->used[k] = 0; }
static void sv_remove_min(sorted_vec *w) { const orc_int k = sv_index_min(w); if (k >= 0) { w->used[k] = 0; sv_pop(w); } }
static void sv_add(sorted_vec *w, orc_int pos, double val) { sv_insert(w, pos, val + (w->used[pos] ? w->key[pos] : 0.0)); }

/* matrix_sparse::diagonally_dominant_symmetric_move_to_corner_improved, sparse_implementation.h:4967-5036 (ROW matrix).
 * Returns 0 when the first phase rejects an index: the reference then re-uses its sorted_vector through resize() (:5014), which
 * refills the multimap but leaves the `used` flags of the first phase (arrays_implementation.h:55-63: vector<bool>::resize keeps
 * existing elements) -- remove() and add() no longer erase, index_min() returns the refilled zero entries in index order, and the
 * result is not a permutation (indices repeat); what follows is undefined (the reference aborts on some inputs).  Nothing to pin. */
static int mat_dd_move_corner(const orc_mat *M, orc_int *P)
{
    const orc_int n = M->n;
    const orc_int *ptr = M->ptr, *idx = M->idx;
    const double *val = M->val;
    orc_int i, j, counter = 0;
    orc_mat T;
    sorted_vec w;
    orc_int *unused = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));
    double *row_gap = (double *)malloc(sizeof(double) * ((size_t)n + 1)), *col_gap = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    for (i = 0; i < n; ++i) { row_gap[i] = 2.0; col_gap[i] = 2.0; P[i] = i; }
    sv_init(&w, n);
    mat_change_orientation(M, &T);
    for (i = 0; i < n; ++i) {
        const orc_int cur = sv_index_min(&w);
        int acceptable = (row_gap[cur] >= 0) && (col_gap[cur] >= 0);
        j = ptr[cur];
        while (acceptable && j < ptr[cur + 1]) {
            if (unused[idx[j]] == 1) acceptable = acceptable && (fabs(val[j]) <= col_gap[idx[j]]);
            j++;
        }
        j = T.ptr[cur];
        while (acceptable && j < T.ptr[cur + 1]) {
            if (unused[T.idx[j]] == 1) acceptable = acceptable && (fabs(T.val[j]) <= row_gap[T.idx[j]]);
            j++;
        }
        if (acceptable) {
            P[counter] = cur;
            sv_remove_min(&w);
            unused[cur] = 1;
            for (j = ptr[cur]; j < ptr[cur + 1]; ++j) {
                if (unused[idx[j]] == 0) sv_add(&w, idx[j], fabs(val[j]));
                col_gap[idx[j]] -= fabs(val[j]);
            }
            for (j = T.ptr[cur]; j < T.ptr[cur + 1]; ++j) {
                if (unused[T.idx[j]] == 0) sv_add(&w, T.idx[j], fabs(T.val[j]));
                row_gap[T.idx[j]] -= fabs(T.val[j]);
            }
            counter++;
        } else {
            unused[cur] = -1;
            sv_remove_min(&w);
        }
    }
    if (counter < n) { sv_free(&w); orc_free_mat(&T); free(unused); free(row_gap); free(col_gap); return 0; }
    /* the rejected indices, :5013-5035 (with a working container; never reached, see above) */
    sv_resize(&w, n);
    for (i = 0; i < counter; ++i) sv_remove(&w, P[i]);
    for (i = 0; i < n; ++i)
        if (unused[i] == -1) {
            for (j = ptr[i]; j < ptr[i + 1]; ++j) if (unused[idx[j]] != -1) sv_add(&w, i, fabs(val[j]));
            for (j = T.ptr[i]; j < T.ptr[i + 1]; ++j) if (unused[T.idx[j]] != -1) sv_add(&w, i, fabs(T.val[j]));
        }
    for (i = counter; i < n; ++i) {
        P[i] = sv_index_min(&w);
        sv_remove_min(&w);
        unused[P[i]] = 0;
        for (j = ptr[P[i]]; j < ptr[P[i] + 1]; ++j) if (unused[idx[j]] == -1) sv_add(&w, idx[j], fabs(val[j]));
        for (j = T.ptr[P[i]]; j < T.ptr[P[i] + 1]; ++j) if (unused[T.idx[j]] == -1) sv_add(&w, T.idx[j], fabs(T.val[j]));
    }
    sv_free(&w); orc_free_mat(&T); free(unused); free(row_gap); free(col_gap);
    return 1;
}

/* vector_dense<Integer>::quicksort(index_list&, left, right): the same algorithm as vec_quicksort on integer data (column_perm,
 * pmwm_implementation.h:539-560) */
static void ivec_quicksort(orc_int *data, orc_int *list, orc_int left, orc_int right)
{
    orc_int i, j, m;
    if (left < right) {
        m = data[left]; i = left; j = right;
        while (i <= j) {
            while (data[i] < m) i++;
            while (data[j] > m) j--;
            if (i <= j) {
                const orc_int t = data[i], u = list[i];
                data[i] = data[j]; data[j] = t; list[i] = list[j]; list[j] = u;
                i++; j--;
            }
        }
        ivec_quicksort(data, list, left, j);
        ivec_quicksort(data, list, i, right);
    }
}

/* matrix_sparse::preprocess, :5214-5460, for the steps of orc_ml_params.preprocessing; A is ROW storage and is replaced by the
 * preprocessed matrix.  P, Q, invP, invQ, Drow, Dcol: n each. */
static int mat_preprocess(orc_mat *A, const orc_ml_params *IP, orc_int *P, orc_int *Q, orc_int *invP, orc_int *invQ, double *Drow,
                          double *Dcol, orc_int *bad_at)
{
    const orc_int n = A->n;
    orc_int i, j;
    int s;
    double *D = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    orc_int *p1 = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1)), *p2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1)),
            *ip1 = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1)), *ip2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
    int rc = ORC_OK;
    /* preprocess() keeps ONE set of work permutations p1, p2, ip1, ip2 for all steps of a call (:5217-5218), and two of the steps only
     * resize them without re-initialising (std::vector::resize keeps existing elements): after a step that filled p1, find_pmwm starts with
     * every column "matched" (mate_col = p1, pmwm_implementation.h:411), finds no augmenting path and returns the identity with unit
     * scalings (:460-471); after a PQ step, a second one finds every row and column "taken" (:4580-4581, :4613) and returns the same
     * permutations again.  Reproduced as the reference behaves. */
    int p1_filled = 0, ip_filled = 0;
    *bad_at = n;
    for (i = 0; i < n; ++i) { P[i] = Q[i] = invP[i] = invQ[i] = i; Drow[i] = 1.0; Dcol[i] = 1.0; }
    for (s = 0; s < IP->n_preprocessing && rc == ORC_OK; ++s) {
        switch (IP->preprocessing[s]) {
        case ORC_PRE_NORMALIZE_COLUMNS:                                        /* :5241-5246, normalize_columns :3306-3309 */
            for (i = 0; i < n; ++i) D[i] = 0.0;
            for (j = 0; j < A->ptr[n]; ++j) D[A->idx[j]] += A->val[j] * A->val[j];       /* norm2_of_dim1, :820-833 */
            for (i = 0; i < n; ++i) D[i] = sqrt(D[i]);
            for (j = 0; j < A->ptr[n]; ++j) A->val[j] /= D[A->idx[j]];                   /* inverse_scale(COLUMN), :3279-3282 */
            vec_permute(D, invQ, n);
            for (i = 0; i < n; ++i) Dcol[i] *= D[i];
            *bad_at = n;
            break;
        case ORC_PRE_NORMALIZE_ROWS:                                           /* :5247-5252 */
            for (i = 0; i < n; ++i) {
                D[i] = 0.0;
                for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) D[i] += A->val[j] * A->val[j];
            }
            for (i = 0; i < n; ++i) D[i] = sqrt(D[i]);
            for (i = 0; i < n; ++i)
                for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) A->val[j] /= D[i];
            vec_permute(D, invP, n);
            for (i = 0; i < n; ++i) Drow[i] *= D[i];
            *bad_at = n;
            break;
        case ORC_PRE_PQ_ORDERING:                                              /* :5264-5275 */
            if (!ip_filled) *bad_at = mat_ddPQ(A, ip1, ip2, IP->pq_threshold);
            else *bad_at = 0;                                                  /* count = -1: pos + 1 */
            ip_filled = 1; p1_filled = 1;
            perm_invert(p1, ip1, n);
            perm_invert(p2, ip2, n);
            mat_permute_rows_cols(A, p1, ip2);
            perm_compose_right(P, p1, n);
            perm_compose_right(Q, p2, n);
            perm_invert(invP, P, n);
            perm_invert(invQ, Q, n);
            break;
        case ORC_PRE_MAX_WEIGHTED_MATCHING_ORDERING: {                         /* :5276-5292 */
            double *D2 = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
            if (!p1_filled) {
                orc_int *mr = (orc_int *)malloc(sizeof(orc_int) * (size_t)(n > 0 ? n : 1));
                (void)mat_pmwm(A, mr, p1, D, D2);                              /* maximal_weight_inverse_scales(p1, D1, D2): find_pmwm(A, invP, P, D1, D2), :5731-5734 */
                free(mr);
            } else {
                for (i = 0; i < n; ++i) { p1[i] = i; D[i] = 1.0; D2[i] = 1.0; }
            }
            p1_filled = 1;
            for (i = 0; i < n; ++i)
                for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) A->val[j] /= D[i];                    /* inverse_scale(D1, ROW) */
            for (j = 0; j < A->ptr[n]; ++j) A->val[j] /= D2[A->idx[j]];                           /* inverse_scale(D2, COLUMN) */
            for (i = 0; i < n; ++i) ip2[i] = i;
            mat_permute_rows_cols(A, p1, ip2);                                                    /* permute(p1, ROW), :5520-5522, :5463-5478 */
            vec_permute(D, invP, n);
            for (i = 0; i < n; ++i) Drow[i] *= D[i];
            vec_permute(D2, invQ, n);
            for (i = 0; i < n; ++i) Dcol[i] *= D2[i];
            perm_compose_right(P, p1, n);
            perm_invert(invP, P, n);
            free(D2);
            *bad_at = n;
            break;
        }
        case ORC_PRE_DD_SYMM_MOVE_CORNER_ORDERING_IM:                          /* :5441-5450 */
            if (!mat_dd_move_corner(A, p1)) { rc = ORC_ERR_UNSUPPORTED; break; }
            p1_filled = 1;
            perm_invert(ip1, p1, n);
            mat_permute_rows_cols(A, p1, ip1);                                 /* permute(p1, p1), :5564-5568 -> :5550-5561 */
            perm_compose_right(P, p1, n);
            perm_compose_right(Q, p1, n);
            perm_invert(invP, P, n);
            perm_invert(invQ, Q, n);
            *bad_at = n;
            break;
        case ORC_PRE_UNIT_OR_ZERO_DIAGONAL_SCALING:                            /* :5309-5314, unit_or_zero_diagonal :5172-5178 */
            for (i = 0; i < n; ++i) {
                D[i] = 1.0;
                for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) if (A->idx[j] == i && A->val[j] != 0.0) D[i] = A->val[j];
            }
            for (i = 0; i < n; ++i)
                for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) A->val[j] /= D[i];
            for (i = 0; i < n; ++i) Drow[i] *= D[i];
            *bad_at = n;
            break;
        case ORC_PRE_SPARSE_FIRST_ORDERING: {                                  /* :5293-5299, column_perm pmwm_implementation.h:539-560 */
            orc_int *cnt = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));
            for (j = 0; j < A->ptr[n]; ++j) cnt[A->idx[j]]++;
            for (i = 0; i < n; ++i) p2[i] = i;
            ivec_quicksort(cnt, p2, 0, n - 1);
            if (n > 0 && cnt[0] == 0) for (i = 0; i < n; ++i) p2[i] = i;       /* an empty column: fill_identity */
            free(cnt);
            perm_invert(ip2, p2, n);
            for (i = 0; i < n; ++i) p1[i] = i;
            mat_permute_rows_cols(A, p1, ip2);                                 /* permute(p2, COLUMN): indices relabelled by the inverse, rows re-sorted, :5497-5512 */
            perm_compose_right(Q, p2, n);
            perm_invert(invQ, Q, n);
            *bad_at = n;
            break;
        }
        case ORC_PRE_SYMM_PQ: {                                                /* :5352-5360 */
            /* sym_ddPQ, :4926-4940: rows by increasing (sum of |a_ij|) * (entries of the row); the reference's own quicksort orders ties */
            for (i = 0; i < n; ++i) {
                D[i] = 0.0;
                for (j = A->ptr[i]; j < A->ptr[i + 1]; ++j) D[i] += fabs(A->val[j]);
            }
            for (i = 0; i < n; ++i) D[i] *= (double)(A->ptr[i + 1] - A->ptr[i]);
            for (i = 0; i < n; ++i) p1[i] = i;
            vec_quicksort(D, p1, 0, n - 1);
            p1_filled = 1;
            perm_invert(ip1, p1, n);
            mat_permute_rows_cols(A, p1, ip1);                                 /* permute(p1, p1), :5564-5568 -> :5550-5561 */
            perm_compose_right(P, p1, n);
            perm_compose_right(Q, p1, n);
            perm_invert(invP, P, n);
            perm_invert(invQ, Q, n);
            *bad_at = n;
            break;
        }
        default:
            rc = ORC_ERR_UNSUPPORTED;
        }
    }
    free(D); free(p1); free(p2); free(ip1); free(ip2);
    return rc;
}

/* vector_dense<T>::sort(list, left, right, m), sparse_implementation.h:507-563: partitions so that the m largest values stand at the end
 * (the selection algorithm of Numerical Recipes; which of several equal values end up there is defined by it); the list follows */
static void vec_select_largest(double *data, orc_int *list, orc_int left, orc_int right, orc_int m)
{
    orc_int i, j, mid, a_list;
    const orc_int k = right - m + 1;
#define SW(x, y) do { const double t_ = data[x]; const orc_int u_ = list[x]; data[x] = data[y]; data[y] = t_; list[x] = list[y]; list[y] = u_; } while (0)
    for (;;) {
        if (right <= left + 1) {
            if (right == left + 1 && data[right] < data[left]) SW(left, right);
            break;
        } else {
            double a;
            mid = (left + right) / 2;
            SW(mid, left + 1);
            if (data[left] > data[right]) SW(left, right);
            if (data[left + 1] > data[right]) SW(left + 1, right);
            if (data[left] > data[left + 1]) SW(left, left + 1);
            i = left + 1; j = right;
            a = data[left + 1]; a_list = list[left + 1];
            for (;;) {
                do i++; while (data[i] < a);
                do j--; while (data[j] > a);
                if (j < i) break;
                SW(i, j);
            }
            data[left + 1] = data[j]; list[left + 1] = list[j];
            data[j] = a; list[j] = a_list;
            if (j >= k) right = j - 1;
            if (j <= k) left = i;
        }
    }
#undef SW
}

/* the common tail of the two selections below: more candidates than n => the n largest (by vec_select_largest), then by index */
static orc_int keep_largest_sorted(double *key, orc_int *list, orc_int cnt, orc_int n)
{
    if (cnt > n) {
        const orc_int offset = cnt - n;
        vec_select_largest(key, list, 0, cnt - 1, n);
        memmove(list, list + offset, sizeof(orc_int) * (size_t)n);
        cnt = n;
    }
    qsort(list, (size_t)cnt, sizeof(orc_int), cmp_idx_pair);
    return cnt;
}

/* vector_sparse_dynamic::take_single_weight_largest_elements_by_abs_value_with_threshold, :1360-1415, WEIGHTED_DROPPING branch: the indices
 * in [from, to) whose weight * |value| >= tau -- at most n of them, the largest products --, ascending */
static orc_int take_single_weight(const wvec *v, orc_int *list, double weight, orc_int n, double tau, orc_int from, orc_int to)
{
    orc_int i, cnt = 0;
    double *key = (double *)malloc(sizeof(double) * (size_t)(v->nnz > 0 ? v->nnz : 1));
    for (i = 0; i < v->nnz; ++i) {
        const double product = weight * fabs(v->data[i]);
        if (from <= v->pointer[i] && v->pointer[i] < to && product >= tau) { key[cnt] = product; list[cnt++] = v->pointer[i]; }
    }
    cnt = keep_largest_sorted(key, list, cnt, n);
    free(key);
    return cnt;
}

/* vector_sparse_dynamic::take_largest_elements_by_abs_value_with_threshold, :1322-1357 */
static orc_int take_largest(const wvec *v, orc_int *list, orc_int n, double tau, orc_int from, orc_int to)
{
    orc_int i, cnt = 0;
    double z = 0.0, norm;
    double *key = (double *)malloc(sizeof(double) * (size_t)(v->nnz > 0 ? v->nnz : 1));
    for (i = 0; i < v->nnz; ++i)
        if (from <= v->pointer[i] && v->pointer[i] < to) z += v->data[i] * v->data[i];
    norm = sqrt(z);
    for (i = 0; i < v->nnz; ++i)
        if (from <= v->pointer[i] && v->pointer[i] < to && fabs(v->data[i]) > norm * tau) { key[cnt] = fabs(v->data[i]); list[cnt++] = v->pointer[i]; }
    cnt = keep_largest_sorted(key, list, cnt, n);
    free(key);
    return cnt;
}

static double wv_norm1(const wvec *v)                                         /* :1074-1078 */
{
    double z = 0.0;
    orc_int i;
    for (i = 0; i < v->nnz; ++i) z += fabs(v->data[i]);
    return z;
}

static double wv_norm2(const wvec *v)                                         /* :1080-1085 */
{
    double z = 0.0;
    orc_int i;
    for (i = 0; i < v->nnz; ++i) z += v->data[i] * v->data[i];
    return sqrt(z);
}

/* iluplusplus_precond_parameter::combine, parameters_implementation.h:526-534 (max is std::max) */
static double ml_combine(const orc_ml_params *IP, double x, double y)
{
    switch (IP->combine_factor) {
    case 1: return x + y;
    case 2: return x * y;
    case 3: { const double m = x < y ? y : x; return IP->min_weight < m ? m : IP->min_weight; }
    default: return x < y ? y : x;
    }
}

/* the weight of a row of U (or column of L) for take_single_weight, ILUCDP.hpp:1719-1728 / :1896-1906: `own` is the vector that is dropped,
 * `other` the one of the other factor, dinv the Dinv[k] of that moment */
static double ml_weight(const orc_ml_params *IP, const wvec *own, const wvec *other, double dinv, double inv_estimate, double accumulated)
{
    double w = IP->neutral_element;
    if (IP->drop_rules & ORC_DROP_STANDARD) { double norm = wv_norm2(own); if (norm == 0.0) norm = 1e-16; w = ml_combine(IP, w, IP->weight_standard_drop / norm); }
    if (IP->drop_rules & ORC_DROP_STANDARD2) w = ml_combine(IP, w, IP->weight_standard_drop2);
    if (IP->drop_rules & ORC_DROP_INVERSE) w = ml_combine(IP, w, IP->weight_inverse_drop * inv_estimate);     /* :725 / :920, :1721 / :1894 */
    if (IP->drop_rules & ORC_DROP_WEIGHTED) w = ml_combine(IP, w, IP->weight_weighted_drop * accumulated);     /* :726 / :921, :1722 / :1895 */
    if (IP->drop_rules & ORC_DROP_ERR_PROP) w = ml_combine(IP, w, IP->weight_err_prop_drop * wv_norm1(other));
    if (IP->drop_rules & ORC_DROP_ERR_PROP2) w = ml_combine(IP, w, IP->weight_err_prop_drop2 * wv_norm1(other) / fabs(dinv));
    if (IP->drop_rules & ORC_DROP_PIVOT) w = ml_combine(IP, w, IP->weight_pivot_drop * fabs(dinv));
    if (IP->scale_weight_invdiag) w *= fabs(dinv);
    return w;
}

/* Inverse-based dropping (Bollhoefer's estimate of the growth of the inverse factors), ILUCDP.hpp:680-713 (row of U) / :882-916 (column of
 * L); partialILUC :1676-1710 / :1856-1892: with v the scaled working vector of step k and pk the index its pivot stands for, x and y are two
 * estimates of |e_pk^T U^-1| built from running products vx, vy over the steps IN THEIR ORDER -- x maximises the 1-norm of the updated
 * products, y counts entries that grow / shrink by a factor of two.  Returns max(|x[pk]|, |y[pk]|), the factor of the dropping weight. */
static double inv_update(const wvec *v, orc_int k, orc_int pk, double *x, double *y, double *vx, double *vy)
{
    orc_int j;
    if (k == 0) {
        x[pk] = 1.0; y[pk] = 1.0;
        for (j = 0; j < v->nnz; ++j) vy[v->pointer[j]] = vx[v->pointer[j]] = v->data[j];
    } else {
        const double xplus = 1.0 - vx[pk], xminus = -1.0 - vx[pk], yplus = 1.0 - vy[pk], yminus = -1.0 - vy[pk];
        double nuplus = 0.0, numinus = 0.0;
        orc_int nplus = 0, nminus = 0;
        for (j = 0; j < v->nnz; ++j) nuplus += fabs(vx[v->pointer[j]] + v->data[j] * xplus);
        for (j = 0; j < v->nnz; ++j) numinus += fabs(vx[v->pointer[j]] + v->data[j] * xminus);
        x[pk] = nuplus > numinus ? xplus : xminus;
        for (j = 0; j < v->nnz; ++j) vx[v->pointer[j]] += v->data[j] * x[pk];
        x[pk] = fabs(xplus) < fabs(xminus) ? fabs(xminus) : fabs(xplus);                       /* max(|xplus|, |xminus|): std::max(a, b) = a < b ? b : a */
        for (j = 0; j < v->nnz; ++j) {
            const double vi = vy[v->pointer[j]];
            const double tp = fabs(vi + v->data[j] * yplus), tm = fabs(vi + v->data[j] * yminus);
            const double lim = 2.0 * fabs(vi) < 0.5 ? 0.5 : 2.0 * fabs(vi);                   /* max(2|vi|, 0.5) */
            if (tp > lim) nplus++;
            if ((2.0 * tp < 0.5 ? 0.5 : 2.0 * tp) < fabs(vi)) nplus--;
            if (tm > lim) nminus++;
            if ((2.0 * tm < 0.5 ? 0.5 : 2.0 * tm) < fabs(vi)) nminus--;
        }
        y[pk] = nplus > nminus ? yplus : yminus;
        for (j = 0; j < v->nnz; ++j) vy[v->pointer[j]] += v->data[j] * y[pk];
        y[pk] = fabs(yplus) < fabs(yminus) ? fabs(yminus) : fabs(yplus);
    }
    return fabs(x[pk]) < fabs(y[pk]) ? fabs(y[pk]) : fabs(x[pk]);
}

/* matrix_sparse::partialILUC, ILUCDP.hpp:1405-2231, for the precon_parameter 10 family: err-prop dropping (weight of a row of U = the
 * 1-norm of the scaled column of L and vice versa), DROP_TYPE 0, no improved Schur complement, unbounded fill, the level ends at
 * the first pivot smaller than MIN_PIVOT.  L: COLUMN, U: ROW (both with the 1 first), Dinv, Anew: the Schur complement (ROW). */
static int partial_iluc(const orc_mat *Arow, const orc_ml_params *IP, int force_finish, double threshold, orc_mat *L, orc_mat *U,
                        double *Dinv, orc_mat *Anew, orc_int *zero_pivots)
{
    const orc_int n = Arow->n;
    const orc_int *ptr = Arow->ptr, *idx = Arow->idx;
    const double *val = Arow->val;
    orc_int k, j, h, x, capL, capU, capA = 0;
    orc_int last_row_to_eliminate = n - 1, n_Anew = 0;
    orc_int max_fill_in = IP->max_fill_in > 0 ? IP->max_fill_in : n;       /* :1440-1447: MAX_FILLIN_IS_INF => n; clamped to [1, n] */
    int eliminate = 1;
    double pivot = 0.0, *inv = NULL, *wts = NULL;
    orc_int *firstU, *listU, *firstL, *listL, *listA, *headA, *firstA, *list_L, *list_U;
    wvec z, w;
    if (max_fill_in < 1) max_fill_in = 1;
    if (max_fill_in > n) max_fill_in = n;
    *zero_pivots = 0;
    capL = capU = ptr[n] + n + 16;
    mat_init(L, n, capL, 0);
    mat_init(U, n, capU, 1);
    mat_init(Anew, 0, 1, 1);
    for (k = 0; k < n; ++k) Dinv[k] = 1.0;
    firstU = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int)); listU = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    firstL = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int)); listL = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    listA = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int)); headA = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    firstA = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));
    list_L = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)n + 16)); list_U = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)n + 16));
    wv_init(&z, n, 0); wv_init(&w, n, 0);
    if (IP->drop_rules & ORC_DROP_INVERSE) inv = (double *)calloc(8 * (size_t)n + 1, sizeof(double));     /* xU yU vxU vyU xL yL vxL vyL, :1537-1546 */
    if (IP->drop_rules & (ORC_DROP_WEIGHTED | ORC_DROP_WEIGHTED2)) {                                       /* weightsU, weightsL, :1547-1550 */
        wts = (double *)malloc(sizeof(double) * (2 * (size_t)n + 1));
        for (k = 0; k < 2 * n; ++k) wts[k] = IP->init_weights_lu;
    }
    initialize_sparse_matrix_fields(n, ptr, idx, listA, headA, firstA);
    for (k = 0; k < n; ++k) { listL[k] = -1; listU[k] = -1; }

    for (k = 0; k < n; ++k) {
        orc_int nL = 0, nU;
        double weightL, weightU, invU = 0.0, invL = 0.0;
        wv_zero_reset(&z);                                                     /* (2.) :1575-1583 */
        for (j = firstA[k]; j < ptr[k + 1]; ++j) z.data[wv_slot(&z, idx[j])] = val[j];
        for (h = listL[k]; h != -1; h = listL[h]) {                            /* (3.) :1589-1602 */
            for (j = firstU[h]; j < U->ptr[h + 1]; ++j) {
                x = wv_slot(&z, U->idx[j]);
                z.data[x] -= L->val[firstL[h]] / Dinv[h] * U->val[j];
            }
        }
        /* the level ends at a small pivot, :1619-1636 (z[k] inserts the slot if the diagonal is missing) */
        if (eliminate && !force_finish && (double)k > IP->min_elim_factor * (double)n && IP->small_pivot_terminates
            && fabs(z.data[wv_slot(&z, k)]) < IP->min_pivot) {
            eliminate = 0;
            threshold *= IP->threshold_shift_schur;
            last_row_to_eliminate = k - 1;
            n_Anew = n - k;
            orc_free_mat(Anew);
            capA = ptr[n] + 16;
            mat_init(Anew, n_Anew, capA, 1);
        }
        if (eliminate) {                                                       /* :1637-1642 */
            x = wv_slot(&z, k);
            pivot = z.data[x];
            Dinv[k] = 1.0 / z.data[x];
            for (j = 0; j < z.nnz; ++j) z.data[j] *= Dinv[k];
            z.data[wv_slot(&z, k)] = 0.0;
        }
        if (wts) for (j = 0; j < z.nnz; ++j) wts[z.pointer[j]] += fabs(z.data[j]);         /* weightsU, :1634-1636 */
        wv_zero_reset(&w);                                                     /* (8.) :1651-1675 */
        if (eliminate) {
            for (h = headA[k]; h != -1; h = listA[h])
                if (h > k) w.data[wv_slot(&w, h)] = val[firstA[h]];
            for (h = listU[k]; h != -1; h = listU[h]) {
                for (j = firstL[h]; j < L->ptr[h + 1]; ++j) {
                    x = wv_slot(&w, L->idx[j]);
                    w.data[x] -= U->val[firstU[h]] / Dinv[h] * L->val[j];
                }
            }
        }
        for (j = 0; j < w.nnz; ++j) w.data[j] *= Dinv[k];                      /* w.scale(Dinv[k]), :1665 */
        if (wts) for (j = 0; j < w.nnz; ++j) wts[n + w.pointer[j]] += fabs(w.data[j]);                                /* weightsL, :1670-1675 */
        if (inv && eliminate) invU = inv_update(&z, k, k, inv, inv + n, inv + 2 * (size_t)n, inv + 3 * (size_t)n);      /* :1676-1710 */
        /* dropping, :1716-1764 */
        if (!eliminate) {
            nU = take_largest(&z, list_U, max_fill_in, threshold, last_row_to_eliminate + 1, n);
        } else {
            weightU = ml_weight(IP, &z, &w, Dinv[k], invU, wts ? wts[k] : 0.0);
            nU = take_single_weight(&z, list_U, weightU, max_fill_in - 1, threshold, k + 1, n);
        }
        /* update U or Anew, :1769-1850 */
        if (eliminate) {
            mat_reserve(U, U->ptr[k] + nU + 1, &capU);
            U->val[U->ptr[k]] = 1.0; U->idx[U->ptr[k]] = k;
            for (j = 0; j < nU; ++j) { U->val[U->ptr[k] + j + 1] = z.data[z.occupancy[list_U[j]]]; U->idx[U->ptr[k] + j + 1] = list_U[j]; }
            U->ptr[k + 1] = U->ptr[k] + nU + 1;
            if (pivot == 0.0) { (*zero_pivots)++; Dinv[k] = 1.0; }
        } else {
            const orc_int k_Anew = k - last_row_to_eliminate - 1;
            mat_reserve(U, U->ptr[k] + 1, &capU);
            mat_reserve(Anew, Anew->ptr[k_Anew] + nU, &capA);
            U->val[U->ptr[k]] = 1.0; Dinv[k] = 1.0; U->idx[U->ptr[k]] = k;
            U->ptr[k + 1] = U->ptr[k] + 1;
            for (j = 0; j < nU; ++j) { Anew->val[Anew->ptr[k_Anew] + j] = z.data[z.occupancy[list_U[j]]]; Anew->idx[Anew->ptr[k_Anew] + j] = list_U[j]; }
            Anew->ptr[k_Anew + 1] = Anew->ptr[k_Anew] + nU;
        }
        /* (12.) L, :1855-1975 */
        if (eliminate) {
            if (inv) invL = inv_update(&w, k, k, inv + 4 * (size_t)n, inv + 5 * (size_t)n, inv + 6 * (size_t)n, inv + 7 * (size_t)n);          /* :1856-1892 */
            weightL = ml_weight(IP, &w, &z, Dinv[k], invL, wts ? wts[n + k] : 0.0);   /* (Dinv[k] after the zero-pivot reset of :1786-1792) */
            nL = take_single_weight(&w, list_L, weightL, max_fill_in - 1, threshold, k + 1, n);
            mat_reserve(L, L->ptr[k] + nL + 1, &capL);
            L->val[L->ptr[k]] = 1.0; L->idx[L->ptr[k]] = k;
            for (j = 0; j < nL; ++j) { L->val[L->ptr[k] + j + 1] = w.data[w.occupancy[list_L[j]]]; L->idx[L->ptr[k] + j + 1] = list_L[j]; }
            L->ptr[k + 1] = L->ptr[k] + nL + 1;
        } else {
            mat_reserve(L, L->ptr[k] + 1, &capL);
            L->val[L->ptr[k]] = 1.0; L->idx[L->ptr[k]] = k;
            L->ptr[k + 1] = L->ptr[k] + 1;
        }
        if (eliminate) {                                                       /* :1977-1983 */
            update_sparse_matrix_fields(k, ptr, idx, listA, headA, firstA);
            update_triangular_fields(k, U->ptr, U->idx, listU, firstU);
        }
        update_triangular_fields(k, L->ptr, L->idx, listL, firstL);
    }
    L->nnz = L->ptr[n]; U->nnz = U->ptr[n];
    mat_compress(L, 0.0);                                                      /* compress(), :2053-2054 */
    mat_compress(U, 0.0);
    if (eliminate) {                                                           /* :2055 */
        orc_free_mat(Anew);
        mat_init(Anew, 0, 1, 1);
    } else {
        Anew->nnz = Anew->ptr[n_Anew];
        if (Anew->nnz > 0) {                                                   /* :2057-2065 */
            mat_compress(Anew, 0.0);
            for (j = 0; j < Anew->nnz; ++j) Anew->idx[j] -= last_row_to_eliminate + 1;
            mat_normal_order(Anew);
        }
    }
    wv_free(&z); wv_free(&w);
    free(firstU); free(listU); free(firstL); free(listL); free(listA); free(headA); free(firstA); free(list_L); free(list_U); free(inv); free(wts);
    return ORC_OK;
}


/* index_list::quicksort_with_inverse, sparse_implementation.h:6074-6100 */
static void perm_quicksort_with_inverse(orc_int *p, orc_int *inv, orc_int left, orc_int right)
{
    orc_int i, j, m, t;
    if (left < right) {
        m = p[left]; i = left; j = right;
        while (i <= j) {
            while (p[i] < m) i++;
            while (p[j] > m) j--;
            if (i <= j) {
                t = inv[p[i]]; inv[p[i]] = inv[p[j]]; inv[p[j]] = t;
                t = p[i]; p[i] = p[j]; p[j] = t;
                i++; j--;
            }
        }
        perm_quicksort_with_inverse(p, inv, left, j);
        perm_quicksort_with_inverse(p, inv, i, right);
    }
}

static void links_reserve(orc_int **link, orc_int **who, orc_int *have, orc_int cap)
{
    if (cap > *have) {
        *link = (orc_int *)realloc(*link, sizeof(orc_int) * (size_t)cap);
        *who = (orc_int *)realloc(*who, sizeof(orc_int) * (size_t)cap);
        *have = cap;
    }
}

/* matrix_sparse::partialILUCDP, ILUCDP.hpp:268-1404 -- the Crout factorisation WITH column pivoting (the largest entry of the working
 * row, by piv_tol) and with rows taken in the order of the number of entries they have in L so far -- for the parameter sets without
 * inverse-based / weighted dropping, DROP_TYPE 0, SCHUR_COMPLEMENT 0, EXTERNAL_FINAL_ROW off, FINAL_ROW_CRIT -1..9.
 * L: COLUMN, U: ROW (1 first), both in the PERMUTED numbering at the end (:1150-1151); perm / permrows: column / row taken at step k. */
static int partial_ilucdp(const orc_mat *Arow, const orc_mat *Acol, const orc_ml_params *IP, int force_finish, double threshold,
                          orc_int bp, orc_int bpr, orc_int epr, orc_mat *L, orc_mat *U, double *Dinv, orc_mat *Anew,
                          orc_int *perm, orc_int *permrows, orc_int *inverse_perm, orc_int *inverse_permrows, orc_int *zero_pivots)
{
    const orc_int n = Arow->n;
    const orc_int nnzA = Acol->ptr[n];
    orc_int k, i, j, h, x, p, t, capL, capU, capA = 0, haveL = 0, haveU = 0;
    orc_int last_row_to_eliminate = n - 1, n_Anew = 0, pos_pivot = -1, selected_row;
    orc_int max_fill_in = IP->max_fill_in > 0 ? IP->max_fill_in : n;       /* :352-355 */
    int eliminate = 1, end_level_now = 0;
    double pivot = 0.0, piv_tol = IP->piv_tol, val_larg_el, *inv = NULL, *wts = NULL;
    orc_int *linkU = NULL, *rowU = NULL, *startU, *linkL = NULL, *colL = NULL, *startL, *list_L, *list_U, *numb, *pnum;
    char *non_pivot, *unused_rows;
    wvec z, w;
    if (max_fill_in < 1) max_fill_in = 1;
    if (max_fill_in > n) max_fill_in = n;
    if (epr < 0) epr = 0;                                                      /* :372-375 */
    if (epr >= n) epr = n - 1;
    if (bpr < 0) bpr = 0;
    if (bpr >= n) bpr = n - 1;
    *zero_pivots = 0;
    capL = capU = nnzA + n + 16;
    mat_init(L, n, capL, 0);
    mat_init(U, n, capU, 1);
    mat_init(Anew, 0, 1, 1);
    links_reserve(&linkL, &colL, &haveL, capL);
    links_reserve(&linkU, &rowU, &haveU, capU);
    startU = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1)); startL = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    list_L = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)n + 16)); list_U = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)n + 16));
    numb = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int)); pnum = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 2));
    non_pivot = (char *)malloc((size_t)n + 1); unused_rows = (char *)malloc((size_t)n + 1);
    for (k = 0; k < n; ++k) {
        Dinv[k] = 1.0; perm[k] = permrows[k] = inverse_perm[k] = inverse_permrows[k] = k;     /* :408-413 */
        non_pivot[k] = 1; unused_rows[k] = 1; startU[k] = -1; startL[k] = -1;
    }
    for (k = 0; k < n + 2; ++k) pnum[k] = epr + 1;                              /* :417, :437 */
    pnum[0] = 0;
    wv_init(&z, n, 0); wv_init(&w, n, 0);
    if (IP->drop_rules & ORC_DROP_INVERSE) inv = (double *)calloc(8 * (size_t)n + 1, sizeof(double));     /* :423-425 */
    if (IP->drop_rules & (ORC_DROP_WEIGHTED | ORC_DROP_WEIGHTED2)) {                                       /* :426-428 */
        wts = (double *)malloc(sizeof(double) * (2 * (size_t)n + 1));
        for (k = 0; k < 2 * n; ++k) wts[k] = IP->init_weights_lu;
    }

    for (k = 0; k < n; ++k) {
        orc_int nL = 0, nU;
        double weightL, weightU, invU = 0.0, invL = 0.0;
        if (IP->begin_total_piv && k == bp) piv_tol = 1.0;                      /* :448 */
        selected_row = permrows[k];                                            /* (2.) :453-466 */
        unused_rows[selected_row] = 0;
        wv_zero_reset(&z);
        for (i = Arow->ptr[selected_row]; i < Arow->ptr[selected_row + 1]; ++i)
            if (non_pivot[Arow->idx[i]]) z.data[wv_slot(&z, Arow->idx[i])] = Arow->val[i];
        for (h = startL[selected_row]; h != -1; ) {                            /* (3.) :472-487 */
            const orc_int c = colL[h];
            const double lv = L->val[h];
            h = linkL[h];
            for (j = U->ptr[c]; j < U->ptr[c + 1]; ++j)
                if (non_pivot[U->idx[j]]) { x = wv_slot(&z, U->idx[j]); z.data[x] -= lv / Dinv[c] * U->val[j]; }
        }
        if (eliminate) {                                                       /* the pivot, :540-558 */
            double mx = 0.0;
            val_larg_el = 0.0; pos_pivot = -1;                                 /* abs_max(pos), sparse_implementation.h:1109-1120 */
            for (i = 0; i < z.nnz; ++i)
                if (fabs(z.data[i]) > mx) { mx = fabs(z.data[i]); val_larg_el = z.data[i]; pos_pivot = z.pointer[i]; }
            if (non_pivot[selected_row]) {
                x = wv_slot(&z, selected_row);
                if (fabs(val_larg_el * piv_tol) > fabs(z.data[x]) && pos_pivot >= 0 && IP->piv_tol > 0) pivot = val_larg_el;
                else { pos_pivot = selected_row; pivot = z.data[x]; }
            } else {
                if (fabs(val_larg_el) > 0.0 && pos_pivot >= 0) pivot = val_larg_el;
                else { pos_pivot = perm[k]; pivot = z.data[wv_slot(&z, pos_pivot)]; }
            }
        }
        if (eliminate && !force_finish && (double)k > IP->min_elim_factor * (double)n && IP->small_pivot_terminates
            && fabs(pivot) < IP->min_pivot) {                                  /* :595-612 */
            eliminate = 0;
            end_level_now = 1;
            threshold *= IP->threshold_shift_schur;
            last_row_to_eliminate = k - 1;
            n_Anew = n - k;
            orc_free_mat(Anew);
            capA = nnzA + 16;
            mat_init(Anew, n_Anew, capA, 1);
        }
        if (eliminate) {                                                       /* :613-629 */
            Dinv[k] = 1.0 / pivot;
            for (j = 0; j < z.nnz; ++j) z.data[j] *= Dinv[k];
            z.data[wv_slot(&z, pos_pivot)] = 0.0;
            p = inverse_perm[pos_pivot];
            t = inverse_perm[perm[k]]; inverse_perm[perm[k]] = inverse_perm[pos_pivot]; inverse_perm[pos_pivot] = t;
            t = perm[k]; perm[k] = perm[p]; perm[p] = t;
            non_pivot[pos_pivot] = 0;
        }
        if (wts) for (j = 0; j < z.nnz; ++j) wts[z.pointer[j]] += fabs(z.data[j]);         /* weightsU, :629-631 */
        wv_zero_reset(&w);                                                     /* :633-651 */
        if (eliminate) {
            const orc_int c = perm[k];
            for (i = Acol->ptr[c]; i < Acol->ptr[c + 1]; ++i)
                if (unused_rows[Acol->idx[i]]) w.data[wv_slot(&w, Acol->idx[i])] = Acol->val[i];
            for (h = startU[c]; h != -1; ) {
                const orc_int r = rowU[h];
                const double uv = U->val[h];
                h = linkU[h];
                for (j = L->ptr[r]; j < L->ptr[r + 1]; ++j)
                    if (unused_rows[L->idx[j]]) { x = wv_slot(&w, L->idx[j]); w.data[x] -= uv / Dinv[r] * L->val[j]; }
            }
        }
        for (j = 0; j < w.nnz; ++j) w.data[j] *= Dinv[k];                      /* :652 */
        if (wts) for (j = 0; j < w.nnz; ++j) wts[n + w.pointer[j]] += fabs(w.data[j]);                                /* weightsL, :670-674 */
        if (inv && eliminate) invU = inv_update(&z, k, perm[k], inv, inv + n, inv + 2 * (size_t)n, inv + 3 * (size_t)n);  /* :679-713 */
        if (!eliminate) nU = take_largest(&z, list_U, max_fill_in, threshold, 0, n);      /* :714-716 */
        else {
            weightU = ml_weight(IP, &z, &w, Dinv[k], invU, wts ? wts[perm[k]] : 0.0);
            nU = take_single_weight(&z, list_U, weightU, max_fill_in - 1, threshold, 0, n);
        }
        if (eliminate) {                                                       /* :761-797 (the list backwards) */
            if (U->ptr[k] + nU + 1 > capU) { mat_reserve(U, U->ptr[k] + nU + 1, &capU); links_reserve(&linkU, &rowU, &haveU, capU); }
            U->val[U->ptr[k]] = 1.0; U->idx[U->ptr[k]] = pos_pivot;
            for (j = 0; j < nU; ++j) {
                const orc_int pos = U->ptr[k] + j + 1, c = list_U[nU - 1 - j];
                U->val[pos] = z.data[z.occupancy[c]]; U->idx[pos] = c;
                linkU[pos] = startU[c]; startU[c] = pos; rowU[pos] = k;
            }
            U->ptr[k + 1] = U->ptr[k] + nU + 1;
            if (pivot == 0.0) { (*zero_pivots)++; Dinv[k] = 1.0; }
        } else {                                                               /* :818-847 */
            const orc_int k_Anew = k - last_row_to_eliminate - 1;
            if (U->ptr[k] + 1 > capU) { mat_reserve(U, U->ptr[k] + 1, &capU); links_reserve(&linkU, &rowU, &haveU, capU); }
            mat_reserve(Anew, Anew->ptr[k_Anew] + nU, &capA);
            U->val[U->ptr[k]] = 1.0; Dinv[k] = 1.0; U->idx[U->ptr[k]] = perm[k];
            U->ptr[k + 1] = U->ptr[k] + 1;
            for (j = 0; j < nU; ++j) {
                const orc_int c = list_U[nU - 1 - j];
                Anew->val[Anew->ptr[k_Anew] + j] = z.data[z.occupancy[c]]; Anew->idx[Anew->ptr[k_Anew] + j] = c;
            }
            Anew->ptr[k_Anew + 1] = Anew->ptr[k_Anew] + nU;
        }
        if (eliminate) {                                                       /* L, :849-1005 */
            if (inv) invL = inv_update(&w, k, selected_row, inv + 4 * (size_t)n, inv + 5 * (size_t)n, inv + 6 * (size_t)n, inv + 7 * (size_t)n);   /* :880-916 */
            weightL = ml_weight(IP, &w, &z, Dinv[k], invL, wts ? wts[n + selected_row] : 0.0);
            nL = take_single_weight(&w, list_L, weightL, max_fill_in, threshold, 0, n);
            if (L->ptr[k] + nL + 1 > capL) { mat_reserve(L, L->ptr[k] + nL + 1, &capL); links_reserve(&linkL, &colL, &haveL, capL); }
            L->val[L->ptr[k]] = 1.0; L->idx[L->ptr[k]] = selected_row;
            for (j = 0; j < nL; ++j) {
                const orc_int pos = L->ptr[k] + j + 1;
                orc_int a, b = list_L[j];
                L->val[pos] = w.data[w.occupancy[b]]; L->idx[pos] = b;
                linkL[pos] = startL[b]; startL[b] = pos; colL[pos] = k;
                if (b >= bpr && b <= epr) {                                    /* the rows by their number of entries in L, :964-970 */
                    b = inverse_permrows[b];
                    a = --pnum[++numb[b]];
                    t = inverse_permrows[permrows[a]]; inverse_permrows[permrows[a]] = inverse_permrows[permrows[b]]; inverse_permrows[permrows[b]] = t;
                    t = permrows[a]; permrows[a] = permrows[b]; permrows[b] = t;
                    t = numb[a]; numb[a] = numb[b]; numb[b] = t;
                }
            }
            if (pnum[numb[k] + 1] == k + 1)                                    /* :980-981 */
                perm_quicksort_with_inverse(permrows, inverse_permrows, pnum[numb[k] + 1], pnum[numb[k] + 2] - 1);
            L->ptr[k + 1] = L->ptr[k] + nL + 1;
        } else {
            if (L->ptr[k] + 1 > capL) { mat_reserve(L, L->ptr[k] + 1, &capL); links_reserve(&linkL, &colL, &haveL, capL); }
            L->val[L->ptr[k]] = 1.0; L->idx[L->ptr[k]] = selected_row;
            L->ptr[k + 1] = L->ptr[k] + 1;
        }
        if (eliminate && !force_finish) {                                      /* the level ends by the fill of L, :1018-1092 */
            if ((double)k > IP->min_elim_factor * (double)n) {
                const double dens = (double)nnzA;
                switch (IP->final_row_crit) {
                case -1: if ((double)numb[k] > (IP->move_level_factor * dens) / (double)n) end_level_now = 1; break;
                case 0: if ((double)numb[k] > (0.5 * dens) / (double)n) end_level_now = 1; break;
                case 1: if ((double)numb[k] > dens / (double)n) end_level_now = 1; break;
                case 2: if ((double)numb[k] > (2.0 * dens) / (double)n) end_level_now = 1; break;
                case 3: if ((double)numb[k] > (4.0 * dens) / (double)n) end_level_now = 1; break;
                case 4: if ((double)numb[k] > (6.0 * dens) / (double)n) end_level_now = 1; break;
                case 5: if (numb[k] > 10) end_level_now = 1; break;
                case 6: if ((double)numb[k] > (1.5 * dens) / (double)n) end_level_now = 1; break;
                case 7: if (wv_norm2(&z) > IP->row_u_max) end_level_now = 1; break;
                case 8: if ((double)numb[k] > (3.0 * dens) / (double)n) end_level_now = 1; break;
                case 9: if ((double)numb[k] > (1.2 * dens) / (double)n) end_level_now = 1; break;
                default: break;
                }
            }
            if (end_level_now) {
                eliminate = 0;
                threshold *= IP->threshold_shift_schur;
                last_row_to_eliminate = k;
                n_Anew = n - k - 1;
                orc_free_mat(Anew);
                capA = nnzA + 16;
                mat_init(Anew, n_Anew, capA, 1);
            }
        }
    }
    L->nnz = L->ptr[n]; U->nnz = U->ptr[n];
    mat_compress(L, 0.0);                                                      /* :1131-1132 */
    mat_compress(U, 0.0);
    if (eliminate) { orc_free_mat(Anew); mat_init(Anew, 0, 1, 1); }
    else {
        Anew->nnz = Anew->ptr[n_Anew];
        if (Anew->nnz > 0) {                                                   /* :1136-1145 */
            mat_compress(Anew, 0.0);
            for (j = 0; j < Anew->nnz; ++j) Anew->idx[j] = inverse_perm[Anew->idx[j]] - last_row_to_eliminate - 1;
            mat_normal_order(Anew);
        }
    }
    for (j = 0; j < L->nnz; ++j) L->idx[j] = inverse_permrows[L->idx[j]];      /* permute(permrows, ROW): against the orientation, :5486-5497 */
    mat_normal_order(L);
    for (j = 0; j < U->nnz; ++j) U->idx[j] = inverse_perm[U->idx[j]];          /* U.permute(perm, COLUMN) */
    mat_normal_order(U);
    wv_free(&z); wv_free(&w);
    free(linkU); free(rowU); free(startU); free(linkL); free(colL); free(startL); free(list_L); free(list_U); free(numb); free(pnum); free(inv); free(wts);
    free(non_pivot); free(unused_rows);
    return ORC_OK;
}

static void ml_level_free(struct orc_ml_level *l)
{
    orc_free_mat(&l->L); orc_free_mat(&l->U);
    free(l->D); free(l->perm_rows); free(l->perm_cols); free(l->inv_perm_rows); free(l->inv_perm_cols); free(l->D_l); free(l->D_r);
}

void orc_ml_free(orc_ml *P)
{
    int i;
    if (!P) return;
    for (i = 0; i < P->nlevels; ++i) ml_level_free(&P->lev[i]);
    free(P->lev);
    free(P);
}

void orc_ml_default_params(orc_ml_params *p)       /* default_parameters (:430-501) + init case 10 (:927-934) + set_PQ */
{
    memset(p, 0, sizeof(*p));
    p->threshold = 0.0;
    p->n_preprocessing = 3;
    p->preprocessing[0] = ORC_PRE_NORMALIZE_COLUMNS; p->preprocessing[1] = ORC_PRE_NORMALIZE_ROWS; p->preprocessing[2] = ORC_PRE_PQ_ORDERING;
    p->pq_threshold = 0.0;
    p->max_levels = 100;
    p->min_ml_size = 0;
    p->small_pivot_terminates = 1;
    p->min_pivot = 1e-2;
    p->min_elim_factor = 0.0;
    p->threshold_shift_schur = 0.0;
    p->vary_threshold_factor = 1.0;
    p->use_final_threshold = 0;
    p->final_threshold = 0.0;
    p->max_fill_in = 0;
    p->drop_rules = ORC_DROP_ERR_PROP;
    p->weight_standard_drop = p->weight_standard_drop2 = p->weight_err_prop_drop = p->weight_err_prop_drop2 = p->weight_pivot_drop = 1.0;
    p->weight_inverse_drop = 1.0; p->weight_weighted_drop = 1.0; p->init_weights_lu = 1.0;
    p->combine_factor = 0;
    p->neutral_element = 0.0;
    p->min_weight = 1.0;
    p->scale_weight_invdiag = 0;
    p->piv_tol = 0.0; p->permute_rows = 0; p->total_piv = 0; p->begin_total_piv = 1;      /* init case 10 */
    p->final_row_crit = -1; p->move_level_factor = 2.0; p->row_u_max = 1.5;
}

/* make_preprocessed_multilevelILUCDP, preconditioner_implementation.h:1350-1665, use_ILUC branch */
int orc_ml_create(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, int is_csr, const orc_ml_params *IP, orc_ml **out)
{
    orc_ml *P;
    orc_mat Ak, Anext, view;
    double tau = IP->threshold;
    orc_int matrix_size, nonzeroes;
    int cap = IP->max_levels > 0 ? IP->max_levels + 1 : 1;
    /* :1376-1382 (EXTERNAL_FINAL_ROW is off here) */
    const int use_iluc = (IP->permute_rows == 0 || IP->permute_rows == 1) && (!IP->begin_total_piv || IP->total_piv == 0) && IP->piv_tol == 0.0;
    *out = NULL;
    view.n = n; view.nnz = ptr[n]; view.ptr = (orc_int *)ptr; view.idx = (orc_int *)idx; view.val = (double *)val; view.is_csr = is_csr;
    if (is_csr) mat_copy(&Ak, &view);
    else mat_change_orientation(&view, &Ak);                                   /* :1388-1391 */
    P = (orc_ml *)calloc(1, sizeof(orc_ml));
    P->n = n;
    P->lev = (struct orc_ml_level *)calloc((size_t)cap, sizeof(struct orc_ml_level));
    matrix_size = Ak.n; nonzeroes = Ak.ptr[Ak.n];
    for (;;) {
        const int in_loop = matrix_size > IP->min_ml_size && P->nlevels < IP->max_levels - 1 && nonzeroes > 0;     /* :1405 */
        struct orc_ml_level *l;
        orc_int bad_at, m;
        int rc;
        if (!in_loop && !(matrix_size > 0)) break;                             /* :1550 */
        l = &P->lev[P->nlevels];
        m = Ak.n;
        l->n = m;
        l->perm_rows = (orc_int *)malloc(sizeof(orc_int) * (size_t)m); l->perm_cols = (orc_int *)malloc(sizeof(orc_int) * (size_t)m);
        l->inv_perm_rows = (orc_int *)malloc(sizeof(orc_int) * (size_t)m); l->inv_perm_cols = (orc_int *)malloc(sizeof(orc_int) * (size_t)m);
        l->D_l = (double *)malloc(sizeof(double) * (size_t)m); l->D_r = (double *)malloc(sizeof(double) * (size_t)m);
        l->D = (double *)malloc(sizeof(double) * (size_t)m);
        rc = mat_preprocess(&Ak, IP, l->perm_rows, l->perm_cols, l->inv_perm_rows, l->inv_perm_cols, l->D_l, l->D_r, &bad_at);
        if (rc != ORC_OK) { P->nlevels++; orc_free_mat(&Ak); orc_ml_free(P); return rc; }
        if (!in_loop && IP->use_final_threshold) tau *= IP->final_threshold;   /* :1580-1581 */
        if (use_iluc) rc = partial_iluc(&Ak, IP, in_loop ? 0 : 1, tau, &l->L, &l->U, l->D, &Anext, &l->zero_pivots);
        else {
            /* the windows of the pivoting and of the row reordering, :1442-1459 (a level of the loop) / :1564-1577 (the last one) */
            const orc_int last_row_to_eliminate = in_loop ? (m - 1) / 2 : m - 1;
            orc_int bp, bpr, epr, i;
            orc_int *pc2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)m), *pr2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)m);
            orc_int *ipc2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)m), *ipr2 = (orc_int *)malloc(sizeof(orc_int) * (size_t)m);
            orc_mat Acol;
            switch (IP->permute_rows) {
            case 0: bpr = 0; epr = 0; break;
            case 1: bpr = bad_at; epr = m - 1; break;
            case 2: bpr = 0; epr = in_loop ? last_row_to_eliminate : m - 1; break;
            default: bpr = 0; epr = m - 1; break;
            }
            switch (IP->total_piv) {
            case 0: bp = m; break;
            case 1: bp = in_loop ? last_row_to_eliminate + 1 : bad_at; break;
            default: bp = 0; break;
            }
            mat_change_orientation(&Ak, &Acol);
            rc = partial_ilucdp(&Ak, &Acol, IP, in_loop ? 0 : 1, tau, bp, bpr, epr, &l->L, &l->U, l->D, &Anext, pc2, pr2, ipc2, ipr2, &l->zero_pivots);
            orc_free_mat(&Acol);
            /* permutation_columns.compose(pc1, pc2), permutation_rows.compose(pr1, pr2) and their inverses, :1516-1521 */
            for (i = 0; i < m; ++i) { ipc2[i] = l->perm_cols[pc2[i]]; ipr2[i] = l->perm_rows[pr2[i]]; }
            for (i = 0; i < m; ++i) { l->perm_cols[i] = ipc2[i]; l->perm_rows[i] = ipr2[i]; }
            perm_invert(l->inv_perm_cols, l->perm_cols, m);
            perm_invert(l->inv_perm_rows, l->perm_rows, m);
            free(pc2); free(pr2); free(ipc2); free(ipr2);
        }
        P->nlevels++;
        if (rc != ORC_OK) { orc_free_mat(&Ak); orc_free_mat(&Anext); orc_ml_free(P); return rc; }
        orc_free_mat(&Ak);
        Ak = Anext;                                                            /* :1532 */
        if (!in_loop) break;
        matrix_size = Ak.n; nonzeroes = Ak.n > 0 ? Ak.ptr[Ak.n] : 0;
        tau *= IP->vary_threshold_factor;
    }
    orc_free_mat(&Ak);
    *out = P;
    return ORC_OK;
}

int orc_ml_levels(const orc_ml *P) { return P->nlevels; }

orc_int orc_ml_total_nnz(const orc_ml *P)          /* preconditioner.h:312, preconditioner_implementation.h:551-569 */
{
    orc_int sum = 0;
    int k;
    for (k = 0; k < P->nlevels; ++k) sum += (P->lev[k].L.nnz - P->lev[k].n) + (P->lev[k].U.nnz - P->lev[k].n) + P->lev[k].n;
    return sum;
}

int orc_ml_level(const orc_ml *P, int k, orc_ml_level_view *v)
{
    const struct orc_ml_level *l;
    if (k < 0 || k >= P->nlevels) return ORC_ERR_UNSUPPORTED;
    l = &P->lev[k];
    v->n = l->n; v->L = l->L; v->U = l->U; v->D = l->D; v->perm_rows = l->perm_rows; v->perm_cols = l->perm_cols;
    v->inv_perm_rows = l->inv_perm_rows; v->inv_perm_cols = l->inv_perm_cols; v->D_l = l->D_l; v->D_r = l->D_r; v->zero_pivots = l->zero_pivots;
    return ORC_OK;
}

/* triangular_solve_with_smaller_matrix (+ _permute_first / _permute_last), sparse_implementation.h:4096-4165: the four loops of
 * orc_trisolve on the LAST m entries of x */
static void tail_permute(double *x, orc_int off, orc_int m, const orc_int *perm)
{
    orc_int i;
    double *w = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
    for (i = 0; i < m; ++i) w[i] = x[off + i];
    for (i = 0; i < m; ++i) x[off + i] = w[perm[i]];
    free(w);
}

void orc_ml_apply(const orc_ml *P, int use, double *x)   /* apply_preconditioner_only :103-111 with left :441-453, right :468-486 */
{
    int i;
    orc_int j;
    const orc_int n = P->n;
    if (use == ORC_ID) {
        for (i = 0; i < P->nlevels; ++i) {
            const struct orc_ml_level *l = &P->lev[i];
            const orc_int off = n - l->n;
            for (j = 0; j < l->n; ++j) x[off + j] /= l->D_l[j];
            tail_permute(x, off, l->n, l->perm_rows);
            orc_trisolve(l->n, l->L.ptr, l->L.idx, l->L.val, 0, ORC_LOWER, ORC_ID, x + off);
        }
        for (i = P->nlevels - 1; i >= 0; --i) {
            const struct orc_ml_level *l = &P->lev[i];
            const orc_int off = n - l->n;
            for (j = 0; j < l->n; ++j) x[off + j] *= l->D[j];
            orc_trisolve(l->n, l->U.ptr, l->U.idx, l->U.val, 1, ORC_UPPER, ORC_ID, x + off);
            tail_permute(x, off, l->n, l->inv_perm_cols);
            for (j = 0; j < l->n; ++j) x[off + j] /= l->D_r[j];
        }
    } else {
        for (i = 0; i < P->nlevels; ++i) {
            const struct orc_ml_level *l = &P->lev[i];
            const orc_int off = n - l->n;
            for (j = 0; j < l->n; ++j) x[off + j] /= l->D_r[j];
            tail_permute(x, off, l->n, l->perm_cols);
            orc_trisolve(l->n, l->U.ptr, l->U.idx, l->U.val, 1, ORC_UPPER, ORC_TRANSPOSE, x + off);
            for (j = 0; j < l->n; ++j) x[off + j] *= l->D[j];
        }
        for (i = P->nlevels - 1; i >= 0; --i) {
            const struct orc_ml_level *l = &P->lev[i];
            const orc_int off = n - l->n;
            orc_trisolve(l->n, l->L.ptr, l->L.idx, l->L.val, 0, ORC_LOWER, ORC_TRANSPOSE, x + off);
            tail_permute(x, off, l->n, l->inv_perm_rows);
            for (j = 0; j < l->n; ++j) x[off + j] /= l->D_l[j];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* ILUCP: Crout ILU with column pivoting                                          */
/* ------------------------------------------------------------------------------------------ */

/* vector_sparse_dynamic::take_largest_elements_by_abs_value_with_threshold_pivot_last, sparse_implementation.h:1151-1216: the entries with
 * |x| > norm2 * tau, at most n of them (the largest, by the reference's selection), in slot order, the PIVOT last -- the largest entry if
 * it beats piv_tol * |x[pivot_position]|, else x[pivot_position] (an empty list if that is 0) */
static orc_int take_largest_pivot_last(const wvec *v, orc_int *list, orc_int n, double tau, orc_int pivot_position, double piv_tol)
{
    orc_int i, cnt = 0, t;
    double val_larg_el = 0.0, norm = 0.0, at_pivot;
    double *key = (double *)malloc(sizeof(double) * (size_t)(v->nnz > 0 ? v->nnz : 1));
    for (i = 0; i < v->nnz; ++i) {
        if (fabs(v->data[i]) > val_larg_el) val_larg_el = fabs(v->data[i]);
        norm += v->data[i] * v->data[i];
    }
    at_pivot = v->occupancy[pivot_position] >= 0 ? v->data[v->occupancy[pivot_position]] : 0.0;      /* read(), :1002-1008 */
    if (val_larg_el * piv_tol > fabs(at_pivot)) {                                 /* pivoting */
        norm = sqrt(norm);
        for (i = 0; i < v->nnz; ++i)
            if (fabs(v->data[i]) > norm * tau) { key[cnt] = fabs(v->data[i]); list[cnt++] = v->pointer[i]; }
        if (cnt > n) {
            const orc_int offset = cnt - n;
            vec_select_largest(key, list, 0, cnt - 1, n);
            memmove(list, list + offset, sizeof(orc_int) * (size_t)n);
            cnt = n;
        } else {
            orc_int pos = 0;
            double mx = 0.0;
            for (i = 0; i < cnt; ++i) if (key[i] > mx) { pos = i; mx = key[i]; }
            if (cnt > 0) { t = list[pos]; list[pos] = list[cnt - 1]; list[cnt - 1] = t; }
        }
    } else {
        if (at_pivot == 0) { free(key); return 0; }
        norm = sqrt(norm);
        for (i = 0; i < v->nnz; ++i)
            if (fabs(v->data[i]) > norm * tau && v->pointer[i] != pivot_position) { key[cnt] = fabs(v->data[i]); list[cnt++] = v->pointer[i]; }
        if (cnt > n - 1) {
            const orc_int offset = cnt - n + 1;
            vec_select_largest(key, list, 0, cnt - 1, n);                          /* (sorted for n, n - 1 of them taken: as there) */
            memmove(list, list + offset, sizeof(orc_int) * (size_t)(n - 1));
            list[n - 1] = pivot_position;
            cnt = n;
        } else {
            list[cnt++] = pivot_position;
        }
    }
    free(key);
    return cnt;
}

/* ILUCP4, ILUC.hpp:212-370, on the major-order view of the arrays (for ROW input: of A^T); U by rows with the pivot FIRST and the
 * original column indices, L by columns with its 1 first; perm[k]: the column taken at step k */
int orc_ilucp(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, orc_int max_fill_in, double threshold, double piv_tol,
              orc_int rp, double mem_factor, orc_mat *L, orc_mat *U, orc_int *perm, orc_int *zero_pivots)
{
    orc_int k, i, j, h, x, p, t, reserved, nU, nL;
    orc_int *firstL, *listL, *listA, *headA, *firstA, *linkU, *rowU, *startU, *inverse_perm, *list_L, *list_U;
    char *non_pivot;
    wvec z, w;
    int rc = ORC_OK;
    if (max_fill_in < 1) max_fill_in = 1;                                        /* :226-227 */
    if (max_fill_in > n) max_fill_in = n;
    {
        long a = (long)max_fill_in * (long)n, b = (long)((orc_int)mem_factor) * (long)ptr[n];      /* (Integer) mem_factor * Acol.non_zeroes(), :229: truncated first */
        long r = a < b ? a : b;
        reserved = (orc_int)(r > 0 ? r : 0);
    }
    *zero_pivots = 0;
    firstL = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int)); listL = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    listA = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int)); headA = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    firstA = (orc_int *)calloc((size_t)n + 1, sizeof(orc_int));
    linkU = (orc_int *)malloc(sizeof(orc_int) * ((size_t)reserved + 1)); rowU = (orc_int *)malloc(sizeof(orc_int) * ((size_t)reserved + 1));
    startU = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1)); inverse_perm = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    list_L = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)n + 16)); list_U = (orc_int *)malloc(sizeof(orc_int) * (2 * (size_t)n + 16));
    non_pivot = (char *)malloc((size_t)n + 1);
    mat_init(U, n, reserved > 0 ? reserved : 1, 1);
    mat_init(L, n, reserved > 0 ? reserved : 1, 0);
    wv_init(&z, n, 0); wv_init(&w, n, 0);
    for (k = 0; k < n; ++k) { perm[k] = inverse_perm[k] = k; non_pivot[k] = 1; listL[k] = -1; startU[k] = -1; }
    initialize_sparse_matrix_fields(n, ptr, idx, listA, headA, firstA);

    for (k = 0; k < n; ++k) {
        if (k == rp) piv_tol = 1.0;                                              /* :247-248 */
        wv_zero_reset(&z);
        for (h = headA[k]; h != -1; h = listA[h])                                /* row k of the matrix, through the column lists */
            if (non_pivot[h]) z.data[wv_slot(&z, h)] = val[firstA[h]];
        for (h = listL[k]; h != -1; h = listL[h])
            for (j = U->ptr[h]; j < U->ptr[h + 1]; ++j)
                if (non_pivot[U->idx[j]]) { x = wv_slot(&z, U->idx[j]); z.data[x] -= L->val[firstL[h]] * U->val[j]; }
        nU = take_largest_pivot_last(&z, list_U, max_fill_in, threshold, perm[k], piv_tol);
        if (nU == 0 && threshold > 0.0) nU = take_largest_pivot_last(&z, list_U, max_fill_in, 0.0, perm[k], piv_tol);
        if (nU == 0) {                                                           /* :281-286 */
            (*zero_pivots)++;
            z.data[wv_slot(&z, perm[k])] = 1.0;
            nU = 1; list_U[0] = perm[k];
        }
        if (U->ptr[k] + nU > reserved) { rc = ORC_ERR_MEMORY; break; }
        U->val[U->ptr[k]] = z.data[z.occupancy[list_U[nU - 1]]];
        U->idx[U->ptr[k]] = list_U[nU - 1];
        for (j = 1; j < nU; ++j) {
            const orc_int pos = U->ptr[k] + j, c = list_U[nU - 1 - j];
            U->val[pos] = z.data[z.occupancy[c]]; U->idx[pos] = c;
            linkU[pos] = startU[c]; startU[c] = pos; rowU[pos] = k;
        }
        U->ptr[k + 1] = U->ptr[k] + nU;
        {
            const orc_int c = U->idx[U->ptr[k]];
            p = inverse_perm[c];
            t = inverse_perm[perm[k]]; inverse_perm[perm[k]] = inverse_perm[c]; inverse_perm[c] = t;
            t = perm[k]; perm[k] = perm[p]; perm[p] = t;
            non_pivot[c] = 0;
        }
        wv_zero_reset(&w);
        for (i = ptr[perm[k]]; i < ptr[perm[k] + 1]; ++i)
            if (idx[i] > k) w.data[wv_slot(&w, idx[i])] = val[i];
        for (h = startU[perm[k]]; h != -1; ) {
            const orc_int r = rowU[h];
            const double uv = U->val[h];
            h = linkU[h];
            for (j = L->ptr[r]; j < L->ptr[r + 1]; ++j) { x = wv_slot(&w, L->idx[j]); w.data[x] -= uv * L->val[j]; }
        }
        nL = take_largest(&w, list_L, max_fill_in - 1, threshold, k + 1, n);
        if (L->ptr[k] + nL + 1 > reserved) { rc = ORC_ERR_MEMORY; break; }
        L->val[L->ptr[k]] = 1.0; L->idx[L->ptr[k]] = k;
        for (j = 0; j < nL; ++j) { L->val[L->ptr[k] + j + 1] = w.data[w.occupancy[list_L[j]]] / U->val[U->ptr[k]]; L->idx[L->ptr[k] + j + 1] = list_L[j]; }
        L->ptr[k + 1] = L->ptr[k] + nL + 1;
        update_sparse_matrix_fields(k, ptr, idx, listA, headA, firstA);
        update_triangular_fields(k, L->ptr, L->idx, listL, firstL);
    }
    wv_free(&z); wv_free(&w);
    free(firstL); free(listL); free(listA); free(headA); free(firstA); free(linkU); free(rowU); free(startU); free(inverse_perm);
    free(list_L); free(list_U); free(non_pivot);
    if (rc != ORC_OK) { orc_free_mat(L); orc_free_mat(U); return rc; }
    L->nnz = L->ptr[n]; U->nnz = U->ptr[n];
    mat_compress(L, 0.0); mat_compress(U, 0.0);                                  /* :365-366 */
    return ORC_OK;
}

/* ILUCPPreconditioner (preconditioner_implementation.h:1117-1147) + apply_preconditioner_only: for COLUMN input L is the left factor
 * (lower triangular) and U the right one (PERMUTED upper triangular, triangular_solve_perm, sparse_implementation.h:4166-4253); for ROW
 * input the factors of A^T change sides.  L, U, perm as orc_ilucp returns them for the major-order view of the input. */
void orc_apply_ilucp(const orc_mat *L, const orc_mat *U, const orc_int *perm, int input_is_csr, int use, double *x)
{
    const orc_int n = L->n;
    orc_int k, j;
    double *y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    /* "forward" = the sweep with the stored rows of U as columns: y[perm[k]] /= pivot; y[rest] -= u * y[perm[k]]; x[k] = y[perm[k]] (:4196-4206);
     * "backward" = rows of U against the unknowns already found: y[k] -= u * x[col]; x[perm[k]] = y[k] / pivot (:4209-4218) */
#define PERM_FORWARD() do { memcpy(y, x, sizeof(double) * (size_t)n); \
        for (k = 0; k < n; ++k) { y[perm[k]] /= U->val[U->ptr[k]]; for (j = U->ptr[k] + 1; j < U->ptr[k + 1]; ++j) y[U->idx[j]] -= U->val[j] * y[perm[k]]; } \
        for (k = 0; k < n; ++k) x[k] = y[perm[k]]; } while (0)
#define PERM_BACKWARD() do { memcpy(y, x, sizeof(double) * (size_t)n); \
        for (k = n - 1; k >= 0; --k) { for (j = U->ptr[k] + 1; j < U->ptr[k + 1]; ++j) y[k] -= U->val[j] * x[U->idx[j]]; x[perm[k]] = y[k] / U->val[U->ptr[k]]; } } while (0)
    if (!input_is_csr) {
        if (use == ORC_ID) { orc_trisolve(n, L->ptr, L->idx, L->val, 0, ORC_LOWER, ORC_ID, x); PERM_BACKWARD(); }
        else { PERM_FORWARD(); orc_trisolve(n, L->ptr, L->idx, L->val, 0, ORC_LOWER, ORC_TRANSPOSE, x); }
    } else {
        /* left = U^T (PERMUTED_LOWER_TRIANGULAR, stored by columns = the rows of U), right = L^T (upper triangular, stored by rows = the columns of L) */
        if (use == ORC_ID) { PERM_FORWARD(); orc_trisolve(n, L->ptr, L->idx, L->val, 1, ORC_UPPER, ORC_ID, x); }
        else { orc_trisolve(n, L->ptr, L->idx, L->val, 1, ORC_UPPER, ORC_TRANSPOSE, x); PERM_BACKWARD(); }
    }
#undef PERM_FORWARD
#undef PERM_BACKWARD
    free(y);
}

/* ------------------------------------------------------------------------------------------ */
/* ILUTP: ILUT with column pivoting                                               */
/* ------------------------------------------------------------------------------------------ */

/* the working row of ILUTP2: vector_sparse_dynamic_enhanced (sparse.h:328-360, sparse_implementation.h:2234-2384) -- slots in insertion
 * order, occupancy by index, and a map sorting key -> slot that is walked in ascending key order while entries are added behind the
 * current key and removed at it.  Keys are positions of a permutation: unique.  Here: keyslot[key] and a min-heap of keys. */
typedef struct { orc_int n, nnz, hlen, ntouched; double *data; orc_int *pointer, *occupancy, *keyslot, *heap, *touched; } evec;
static void ev_init(evec *w, orc_int n)
{
    orc_int i;
    w->n = n; w->nnz = 0; w->hlen = 0; w->ntouched = 0;
    w->touched = (orc_int *)malloc(sizeof(orc_int) * (4 * (size_t)n + 16));
    w->data = (double *)malloc(sizeof(double) * (4 * (size_t)n + 16)); w->pointer = (orc_int *)malloc(sizeof(orc_int) * (4 * (size_t)n + 16));
    w->occupancy = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1)); w->keyslot = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    w->heap = (orc_int *)malloc(sizeof(orc_int) * (4 * (size_t)n + 16));
    for (i = 0; i < n; ++i) { w->occupancy[i] = -1; w->keyslot[i] = -1; }
}
static void ev_free(evec *w) { free(w->data); free(w->pointer); free(w->occupancy); free(w->keyslot); free(w->heap); free(w->touched); }
static void ev_heap_push(evec *w, orc_int key)
{
    orc_int i = w->hlen++, p;
    w->heap[i] = key;
    while (i > 0 && w->heap[p = (i - 1) / 2] > w->heap[i]) { const orc_int t = w->heap[p]; w->heap[p] = w->heap[i]; w->heap[i] = t; i = p; }
}
static void ev_heap_pop(evec *w)
{
    orc_int i = 0, c;
    w->heap[0] = w->heap[--w->hlen];
    for (;;) {
        c = 2 * i + 1;
        if (c >= w->hlen) break;
        if (c + 1 < w->hlen && w->heap[c + 1] < w->heap[c]) c++;
        if (w->heap[i] <= w->heap[c]) break;
        { const orc_int t = w->heap[c]; w->heap[c] = w->heap[i]; w->heap[i] = t; }
        i = c;
    }
}
/* operator()(j, k), :2250-2271: the slot of index j (a new one with sorting key k if j is not occupied) */
static orc_int ev_slot(evec *w, orc_int j, orc_int k)
{
    if (w->occupancy[j] < 0) {
        const orc_int x = w->nnz++;
        w->pointer[x] = j; w->data[x] = 0.0; w->occupancy[j] = x;
        ev_heap_push(w, k);
        w->keyslot[k] = x;
        w->touched[w->ntouched++] = k;
        return x;
    }
    return w->occupancy[j];
}
static void ev_reset(evec *w)       /* zero_reset, :2234-2240 */
{
    orc_int i;
    for (i = 0; i < w->nnz; ++i) w->occupancy[w->pointer[i]] = -1;
    for (i = 0; i < w->ntouched; ++i) w->keyslot[w->touched[i]] = -1;
    w->nnz = 0; w->hlen = 0; w->ntouched = 0;
}

/* the two selections of ILUTP2 (sparse_implementation.h:1943-2033 without, :2036-2160 with the pivot tolerance): entries whose position
 * invperm[index] lies left of `mid` go to L, the others to U, each with its own norm and threshold; at most n_L / n_U of them (the largest,
 * by the reference's partial sort); in U the largest of the kept ones is swapped to the end (it becomes the pivot).  WITH piv_tol: unless
 * the largest entry of the U part beats the diagonal by piv_tol, the diagonal's magnitude is replaced by the norm, which makes it the
 * largest (:2083-2087). */
static void ilutp_select(const evec *v, const orc_int *invperm, orc_int n_L, orc_int n_U, double tau_L, double tau_U, orc_int mid, int with_piv,
                         double piv_tol, orc_int *list_L, orc_int *nL, orc_int *list_U, orc_int *nU)
{
    orc_int i, cL = 0, cU = 0, pos, t;
    double nrmL = 0.0, nrmU = 0.0, larg = 0.0, potpiv = 0.0;
    orc_int pos_pot = -1, keep_diag = -1;
    double *kL = (double *)malloc(sizeof(double) * (size_t)(v->nnz > 0 ? v->nnz : 1)), *kU = (double *)malloc(sizeof(double) * (size_t)(v->nnz > 0 ? v->nnz : 1));
    *nL = *nU = 0;
    if (v->n == 0) { free(kL); free(kU); return; }
    for (i = 0; i < v->nnz; ++i) {
        const double a = fabs(v->data[i]);
        const orc_int p = invperm[v->pointer[i]];
        if (p < mid) nrmL += v->data[i] * v->data[i];
        else {
            nrmU += v->data[i] * v->data[i];
            if (with_piv) {
                if (a > larg) larg = a;
                if (p == mid) { potpiv = a; pos_pot = i; }
            }
        }
    }
    nrmL = sqrt(nrmL); nrmU = sqrt(nrmU);
    /* not pivoting (:2083-2087): the diagonal entry counts as large as the norm -- it is selected and goes to the end as the pivot */
    if (with_piv && !((larg * piv_tol >= potpiv) || (pos_pot < 0))) keep_diag = pos_pot;
    for (i = 0; i < v->nnz; ++i) {
        const double a = i == keep_diag ? nrmU : fabs(v->data[i]);
        if (invperm[v->pointer[i]] < mid) { if (a > nrmL * tau_L) { kL[cL] = a; list_L[cL++] = v->pointer[i]; } }
        else if (a > nrmU * tau_U) { kU[cU] = a; list_U[cU++] = v->pointer[i]; }
    }
    if (cL > n_L) {
        const orc_int offset = cL - n_L;
        vec_select_largest(kL, list_L, 0, cL - 1, n_L);
        memmove(list_L, list_L + offset, sizeof(orc_int) * (size_t)n_L);
        cL = n_L;
    }
    if (cU > 0) {
        if (cU > n_U) {
            const orc_int offset = cU - n_U;
            vec_select_largest(kU, list_U, 0, cU - 1, n_U);
            pos = offset;
            for (i = offset + 1; i < cU; ++i) if (kU[i] > kU[pos]) pos = i;
            t = list_U[pos]; list_U[pos] = list_U[cU - 1]; list_U[cU - 1] = t;
            memmove(list_U, list_U + offset, sizeof(orc_int) * (size_t)n_U);
            cU = n_U;
        } else {
            pos = 0;
            for (i = 1; i < cU; ++i) if (kU[i] > kU[pos]) pos = i;
            t = list_U[pos]; list_U[pos] = list_U[cU - 1]; list_U[cU - 1] = t;
        }
    }
    *nL = cL; *nU = cU;
    free(kL); free(kU);
}

/* ILUTP2, ILUTP.hpp:13-140, on the major-order view of the arrays (for COLUMN input: of A^T): L by rows (its 1 last, columns in the
 * PERMUTED numbering, sorted), U by rows (the pivot first, ORIGINAL column indices, ordered by their permuted position); perm[i]: the
 * column taken at step i.  Errors: ORC_ERR_MEMORY ("ILUTP2: memory reserved was insufficient."), ORC_ERR_ZERO_PIVOT ("encountered zero pivot") */
int orc_ilutp(orc_int n, const orc_int *ptr, const orc_int *idx, const double *val, orc_int max_fill_in, double threshold, double piv_tol,
              orc_int bp, double mem_factor, orc_mat *L, orc_mat *U, orc_int *perm, orc_int *zero_pivots)
{
    orc_int i, j, k, p, t, x, reserved, nL, nU;
    orc_int *inverse_perm, *list_L, *list_U;
    evec w;
    int rc = ORC_OK;
    if (max_fill_in < 1) max_fill_in = 1;
    if (max_fill_in > n) max_fill_in = n;
    {
        long a = (long)max_fill_in * (long)n, b = (long)((orc_int)mem_factor) * (long)ptr[n];      /* (Integer) mem_factor * A.non_zeroes(), :37 */
        long r = a < b ? a : b;
        reserved = (orc_int)(r > 0 ? r : 0);
    }
    *zero_pivots = 0;
    inverse_perm = (orc_int *)malloc(sizeof(orc_int) * ((size_t)n + 1));
    list_L = (orc_int *)malloc(sizeof(orc_int) * (4 * (size_t)n + 16)); list_U = (orc_int *)malloc(sizeof(orc_int) * (4 * (size_t)n + 16));
    mat_init(U, n, reserved > 0 ? reserved : 1, 1);
    mat_init(L, n, reserved > 0 ? reserved : 1, 1);
    ev_init(&w, n);
    for (k = 0; k < n; ++k) perm[k] = inverse_perm[k] = k;
    for (i = 0; i < n; ++i) {
        double norm_wL = 0.0;
        if (i == bp) piv_tol = 1.0;
        for (k = ptr[i]; k < ptr[i + 1]; ++k) {                                  /* :46-52 */
            j = inverse_perm[idx[k]];
            w.data[ev_slot(&w, idx[k], j)] = val[k];
            if (j < i) norm_wL += val[k] * val[k];
        }
        norm_wL = sqrt(norm_wL);
        while (w.hlen > 0 && w.heap[0] < i) {                                    /* :54-67: the entries left of position i, by position */
            k = w.heap[0];
            x = w.keyslot[k];
            if (fabs(w.data[x]) < threshold * norm_wL) {                         /* current_zero_set, :2376-2384 */
                if (w.occupancy[w.pointer[x]] >= 0) { w.data[w.occupancy[w.pointer[x]]] = 0.0; w.occupancy[w.pointer[x]] = -1; }
                w.keyslot[k] = -1; ev_heap_pop(&w);
            } else {
                double wk;
                w.data[x] /= U->val[U->ptr[k]];
                wk = w.data[x];
                ev_heap_pop(&w);
                for (j = U->ptr[k] + 1; j < U->ptr[k + 1]; ++j) {
                    const orc_int y = ev_slot(&w, U->idx[j], inverse_perm[U->idx[j]]);
                    w.data[y] -= wk * U->val[j];
                }
            }
        }
        ilutp_select(&w, inverse_perm, max_fill_in - 1, max_fill_in, threshold, threshold, i, 1, piv_tol, list_L, &nL, list_U, &nU);
        if (nU == 0) {                                                           /* :72-84 */
            if (threshold > 0.0) ilutp_select(&w, inverse_perm, max_fill_in - 1, max_fill_in, threshold, 0.0, i, 0, 0.0, list_L, &nL, list_U, &nU);
            if (nU == 0) {
                (*zero_pivots)++;
                w.data[ev_slot(&w, perm[i], i)] = 1.0;
                nU = 1; list_U[0] = perm[i];
            }
        }
        if (L->ptr[i] + nL + 1 > reserved) { rc = ORC_ERR_MEMORY; break; }
        for (j = 0; j < nL; ++j) {
            const orc_int c = list_L[nL - 1 - j];
            L->val[L->ptr[i] + j] = w.data[w.occupancy[c]];
            L->idx[L->ptr[i] + j] = inverse_perm[c];
        }
        L->val[L->ptr[i] + nL] = 1.0; L->idx[L->ptr[i] + nL] = i;
        L->ptr[i + 1] = L->ptr[i] + nL + 1;
        if (U->ptr[i] + nU > reserved) { rc = ORC_ERR_MEMORY; break; }
        for (j = 0; j < nU; ++j) {
            const orc_int c = list_U[nU - 1 - j];
            U->val[U->ptr[i] + j] = w.data[w.occupancy[c]];
            U->idx[U->ptr[i] + j] = c;
        }
        U->ptr[i + 1] = U->ptr[i] + nU;
        {
            const orc_int c = U->idx[U->ptr[i]];
            p = inverse_perm[c];
            t = inverse_perm[perm[i]]; inverse_perm[perm[i]] = inverse_perm[c]; inverse_perm[c] = t;
            t = perm[i]; perm[i] = perm[p]; perm[p] = t;
        }
        if (U->val[U->ptr[i]] == 0) { rc = ORC_ERR_ZERO_PIVOT; break; }
        ev_reset(&w);
    }
    ev_free(&w);
    free(list_L); free(list_U);
    if (rc != ORC_OK) { free(inverse_perm); orc_free_mat(L); orc_free_mat(U); return rc; }
    L->nnz = L->ptr[n]; U->nnz = U->ptr[n];
    mat_compress(L, 0.0); mat_compress(U, 0.0);                                  /* :131-132 */
    {   /* U.reorder(inverse_perm), sparse_implementation.h:3357-3379: every row by the permuted position of its columns; L.normal_order() */
        orc_int r, maxlen = 0;
        struct pr2 { orc_int key, pos; } *buf;
        orc_int *ti; double *tv;
        for (r = 0; r < n; ++r) if (U->ptr[r + 1] - U->ptr[r] > maxlen) maxlen = U->ptr[r + 1] - U->ptr[r];
        buf = (struct pr2 *)malloc(sizeof(struct pr2) * (size_t)(maxlen > 0 ? maxlen : 1));
        ti = (orc_int *)malloc(sizeof(orc_int) * (size_t)(maxlen > 0 ? maxlen : 1)); tv = (double *)malloc(sizeof(double) * (size_t)(maxlen > 0 ? maxlen : 1));
        for (r = 0; r < n; ++r) {
            const orc_int b0 = U->ptr[r], len = U->ptr[r + 1] - b0;
            for (j = 0; j < len; ++j) { buf[j].key = inverse_perm[U->idx[b0 + j]]; buf[j].pos = j; ti[j] = U->idx[b0 + j]; tv[j] = U->val[b0 + j]; }
            qsort(buf, (size_t)len, sizeof(struct pr2), cmp_idx_pair);
            for (j = 0; j < len; ++j) { U->idx[b0 + j] = ti[buf[j].pos]; U->val[b0 + j] = tv[buf[j].pos]; }
        }
        free(buf); free(ti); free(tv);
    }
    mat_normal_order(L);
    free(inverse_perm);
    return ORC_OK;
}

/* ILUTPPreconditioner (preconditioner_implementation.h:1050-1078) + apply_preconditioner_only: for ROW input L (by rows, its 1 last) is the left
 * factor, U (by rows, PERMUTED upper triangular) the right one; for COLUMN input the factors of A^T change sides. */
void orc_apply_ilutp(const orc_mat *L, const orc_mat *U, const orc_int *perm, int input_is_csr, int use, double *x)
{
    const orc_int n = L->n;
    orc_int k, j;
    double *y = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#define PERM_FORWARD() do { memcpy(y, x, sizeof(double) * (size_t)n); \
        for (k = 0; k < n; ++k) { y[perm[k]] /= U->val[U->ptr[k]]; for (j = U->ptr[k] + 1; j < U->ptr[k + 1]; ++j) y[U->idx[j]] -= U->val[j] * y[perm[k]]; } \
        for (k = 0; k < n; ++k) x[k] = y[perm[k]]; } while (0)
#define PERM_BACKWARD() do { memcpy(y, x, sizeof(double) * (size_t)n); \
        for (k = n - 1; k >= 0; --k) { for (j = U->ptr[k] + 1; j < U->ptr[k + 1]; ++j) y[k] -= U->val[j] * x[U->idx[j]]; x[perm[k]] = y[k] / U->val[U->ptr[k]]; } } while (0)
    if (input_is_csr) {
        if (use == ORC_ID) { orc_trisolve(n, L->ptr, L->idx, L->val, 1, ORC_LOWER, ORC_ID, x); PERM_BACKWARD(); }
        else { PERM_FORWARD(); orc_trisolve(n, L->ptr, L->idx, L->val, 1, ORC_LOWER, ORC_TRANSPOSE, x); }
    } else {
        /* left = U^T (PERMUTED_LOWER_TRIANGULAR, by columns = the rows of U), right = L^T (upper triangular, by columns = the rows of L) */
        if (use == ORC_ID) { PERM_FORWARD(); orc_trisolve(n, L->ptr, L->idx, L->val, 0, ORC_UPPER, ORC_ID, x); }
        else { orc_trisolve(n, L->ptr, L->idx, L->val, 0, ORC_UPPER, ORC_TRANSPOSE, x); PERM_BACKWARD(); }
    }
#undef PERM_FORWARD
#undef PERM_BACKWARD
    free(y);
}
