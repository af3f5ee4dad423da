// ilupp_amd/csrc/piluc_dev.h -- device helpers shared by the two multilevel level kernels (piluc_df.hip, pilucdp.hip)
#pragma once
#include "common.h"

namespace ilupp {

// bitonic sort of N (a power of two) 64-bit keys by one wave, ascending; the array in LDS or, for the global class, in the wave's slice
// of global memory
template <bool kGlobal>
__device__ __forceinline__ void wave_sort_u64(unsigned long long *a, int N, int lane)
{
    for (int k2 = 2; k2 <= N; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < N; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = a[i], y = a[l];
                    const bool up = (i & k2) == 0;
                    if ((x > y) == up) { a[i] = y; a[l] = x; }
                }
            }
            if (kGlobal) __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
}
// vector_dense<T>::sort(list, left, right, m), sparse_implementation.h:507-563: the selection algorithm the reference uses when a row has more
// candidates than it may keep -- afterwards the m largest keys stand at the end; which of several equal keys do is defined by this algorithm,
// so it runs as written, on one lane (rows that exceed a bounded fill are rare)
__device__ inline void select_largest(double *data, int *list, int left, int right, int m)
{
    const int k = right - m + 1;
#define PSW(x, y) do { const double t_ = data[x]; const int u_ = list[x]; data[x] = data[y]; data[y] = t_; list[x] = list[y]; list[y] = u_; } while (0)
    for (;;) {
        if (right <= left + 1) {
            if (right == left + 1 && data[right] < data[left]) PSW(left, right);
            break;
        }
        const int mid = (left + right) / 2;
        PSW(mid, left + 1);
        if (data[left] > data[right]) PSW(left, right);
        if (data[left + 1] > data[right]) PSW(left + 1, right);
        if (data[left] > data[left + 1]) PSW(left, left + 1);
        int i = left + 1, j = right;
        const double a = data[left + 1];
        const int a_list = list[left + 1];
        for (;;) {
            do i++; while (data[i] < a);
            do j--; while (data[j] > a);
            if (j < i) break;
            PSW(i, j);
        }
        data[left + 1] = data[j]; list[left + 1] = list[j];
        data[j] = a; list[j] = a_list;
        if (j >= k) right = j - 1;
        if (j <= k) left = i;
    }
#undef PSW
}


}  // namespace ilupp
