// Ranking metrics of the reference's eval stage on the device (reference src/evl/metric.py:5-35,44-73, which
// goes through pytrec_eval): P_k, recall_k, ndcg_cut_k, map_cut_k, success_k from a ranked top-K list and the CSR truth row, and the
// skill coverage of the top-k experts.  One wave per test instance; integer/index work, latency-bound, nothing here is MFMA-shaped.
#include "../../include/opentf_amd.h"
#include "ntf_device.h"
#include <string>
#include <vector>

namespace ntf {

constexpr int MAX_CUT = 8;
struct Cutoffs { int n; int k[MAX_CUT]; };

// out[i, m*ncut + q], m in {P, recall, ndcg_cut, map_cut, success}
__global__ __launch_bounds__(64) void k_rank_metrics(const int32_t* __restrict__ topk, int K, const int64_t* __restrict__ t_indptr,
                                                     const int32_t* __restrict__ t_indices, const int64_t* __restrict__ rows, Cutoffs cut,
                                                     float* __restrict__ out) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t trow = rows ? rows[i] : i;
    const int64_t tb = t_indptr[trow];
    const int R = (int)(t_indptr[trow + 1] - tb);
    int kmax = 0;
    for (int q = 0; q < cut.n; ++q) kmax = max(kmax, cut.k[q]);
    float hits[MAX_CUT], dcg[MAX_CUT], ap[MAX_CUT];
    for (int q = 0; q < MAX_CUT; ++q) { hits[q] = 0.f; dcg[q] = 0.f; ap[q] = 0.f; }
    int seen_before = 0;  // relevant documents in the ranked positions before this 64-wide window
    for (int base = 0; base < min(kmax, K); base += 64) {
        const int p = base + lane;
        int rel = 0;
        if (p < kmax && p < K) {
            const int c = topk[i * K + p];
            for (int j = 0; j < R; ++j) rel |= (t_indices[tb + j] == c);
        }
        // inclusive prefix count of relevant documents inside the window
        int pre = rel;
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(pre, o, 64); if (lane >= o) pre += v; }
        const int seen = seen_before + pre;
        const float g = rel ? 1.f / log2f((float)(p + 2)) : 0.f;
        const float a = rel ? (float)seen / (float)(p + 1) : 0.f;
        for (int q = 0; q < cut.n; ++q) {
            const bool in = p < cut.k[q];
            hits[q] += wave_reduce_sum(in && rel ? 1.f : 0.f);
            dcg[q] += wave_reduce_sum(in ? g : 0.f);
            ap[q] += wave_reduce_sum(in ? a : 0.f);
        }
        seen_before += __shfl(pre, 63, 64);
    }
    if (lane == 0) {
        const int nc = cut.n;
        for (int q = 0; q < nc; ++q) {
            const int k = cut.k[q];
            float idcg = 0.f;
            for (int j = 0; j < min(R, k); ++j) idcg += 1.f / log2f((float)(j + 2));
            out[i * 5 * nc + 0 * nc + q] = hits[q] / (float)k;
            out[i * 5 * nc + 1 * nc + q] = R ? hits[q] / (float)R : 0.f;
            out[i * 5 * nc + 2 * nc + q] = idcg > 0.f ? dcg[q] / idcg : 0.f;
            out[i * 5 * nc + 3 * nc + q] = R ? ap[q] / (float)R : 0.f;
            out[i * 5 * nc + 4 * nc + q] = hits[q] > 0.f ? 1.f : 0.f;
        }
    }
}

// coverage_k = |skills held by the top-k experts  ∩  required skills| / |required skills|        (metric.py:63-69)
// lane = one required skill: the first ranked position at which an expert holds it (binary search in the expert's sorted skill list)
__global__ __launch_bounds__(64) void k_skill_coverage(const int32_t* __restrict__ topk, int K, const int64_t* __restrict__ s_indptr,
                                                       const int32_t* __restrict__ s_indices, const int64_t* __restrict__ rows,
                                                       const int64_t* __restrict__ c_indptr, const int32_t* __restrict__ c_indices, Cutoffs cut,
                                                       float* __restrict__ out) {
    const int64_t i = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t srow = rows ? rows[i] : i;
    const int64_t sb = s_indptr[srow];
    const int nreq = (int)(s_indptr[srow + 1] - sb);
    int kmax = 0;
    for (int q = 0; q < cut.n; ++q) kmax = max(kmax, cut.k[q]);
    kmax = min(kmax, K);
    float covered[MAX_CUT];
    for (int q = 0; q < MAX_CUT; ++q) covered[q] = 0.f;
    for (int base = 0; base < nreq; base += 64) {
        const int j = base + lane;
        int first = 0x7fffffff;
        if (j < nreq) {
            const int s = s_indices[sb + j];
            for (int p = 0; p < kmax && first == 0x7fffffff; ++p) {
                const int e = topk[i * K + p];
                int64_t lo = c_indptr[e], hi = c_indptr[e + 1];
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (c_indices[mid] < s) lo = mid + 1; else hi = mid; }
                if (lo < c_indptr[e + 1] && c_indices[lo] == s) first = p;
            }
        }
        for (int q = 0; q < cut.n; ++q) covered[q] += wave_reduce_sum(first < cut.k[q] ? 1.f : 0.f);
    }
    if (lane == 0) for (int q = 0; q < cut.n; ++q) out[i * cut.n + q] = covered[q] / (float)nreq;
}

}  // namespace ntf

using namespace ntf;

namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    bool put(const void* host, size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return false;
        return !bytes || hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) == hipSuccess;
    }
};
bool make_cut(const int32_t* cutoffs, int n, Cutoffs& c) {
    if (!cutoffs || n < 1 || n > MAX_CUT) return false;
    c.n = n;
    for (int q = 0; q < n; ++q) { if (cutoffs[q] < 1) return false; c.k[q] = cutoffs[q]; }
    return true;
}
}  // namespace

extern "C" int ntf_rank_metrics(int device, const int32_t* topk_idx, int64_t n, int32_t K, const int64_t* truth_indptr, const int32_t* truth_indices,
                                int64_t n_truth_rows, const int64_t* rows, const int32_t* cutoffs, int32_t n_cut, float* out) {
    Cutoffs c;
    if (!topk_idx || n < 1 || K < 1 || !truth_indptr || n_truth_rows < 1 || !out || !make_cut(cutoffs, n_cut, c)) return NTF_EINVAL;
    if (rows) { for (int64_t i = 0; i < n; ++i) if (rows[i] < 0 || rows[i] >= n_truth_rows) return NTF_EINVAL; } else if (n > n_truth_rows) return NTF_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return NTF_EHIP;
    DevBuf dk, dip, dix, dr, dout;
    const int64_t nnz = truth_indptr[n_truth_rows];
    if (!dk.put(topk_idx, (size_t)n * K * 4) || !dip.put(truth_indptr, (size_t)(n_truth_rows + 1) * 8) || !dix.put(truth_indices, (size_t)nnz * 4) ||
        (rows && !dr.put(rows, (size_t)n * 8)) || !dout.put(nullptr, 0)) return NTF_ENOMEM;
    hipFree(dout.p); dout.p = nullptr;
    if (hipMalloc(&dout.p, (size_t)n * 5 * n_cut * 4) != hipSuccess) return NTF_ENOMEM;
    hipLaunchKernelGGL(k_rank_metrics, dim3((unsigned)n), dim3(64), 0, 0, (const int32_t*)dk.p, K, (const int64_t*)dip.p, (const int32_t*)dix.p,
                       rows ? (const int64_t*)dr.p : nullptr, c, (float*)dout.p);
    if (hipMemcpy(out, dout.p, (size_t)n * 5 * n_cut * 4, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    return NTF_OK;
}

extern "C" int ntf_skill_coverage(int device, const int32_t* topk_idx, int64_t n, int32_t K, const int64_t* skill_indptr, const int32_t* skill_indices,
                                  int64_t n_skill_rows, const int64_t* rows, const int64_t* cov_indptr, const int32_t* cov_indices, int64_t n_experts,
                                  const int32_t* cutoffs, int32_t n_cut, float* out) {
    Cutoffs c;
    if (!topk_idx || n < 1 || K < 1 || !skill_indptr || !cov_indptr || n_skill_rows < 1 || n_experts < 1 || !out || !make_cut(cutoffs, n_cut, c)) return NTF_EINVAL;
    if (rows) { for (int64_t i = 0; i < n; ++i) if (rows[i] < 0 || rows[i] >= n_skill_rows) return NTF_EINVAL; } else if (n > n_skill_rows) return NTF_EINVAL;
    for (int64_t i = 0; i < n * K; ++i) if (topk_idx[i] < 0 || topk_idx[i] >= n_experts) return NTF_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return NTF_EHIP;
    DevBuf dk, sip, six, dr, cip, cix, dout;
    if (!dk.put(topk_idx, (size_t)n * K * 4) || !sip.put(skill_indptr, (size_t)(n_skill_rows + 1) * 8) || !six.put(skill_indices, (size_t)skill_indptr[n_skill_rows] * 4) ||
        (rows && !dr.put(rows, (size_t)n * 8)) || !cip.put(cov_indptr, (size_t)(n_experts + 1) * 8) || !cix.put(cov_indices, (size_t)cov_indptr[n_experts] * 4)) return NTF_ENOMEM;
    if (hipMalloc(&dout.p, (size_t)n * n_cut * 4) != hipSuccess) return NTF_ENOMEM;
    hipLaunchKernelGGL(k_skill_coverage, dim3((unsigned)n), dim3(64), 0, 0, (const int32_t*)dk.p, K, (const int64_t*)sip.p, (const int32_t*)six.p,
                       rows ? (const int64_t*)dr.p : nullptr, (const int64_t*)cip.p, (const int32_t*)cix.p, c, (float*)dout.p);
    if (hipMemcpy(out, dout.p, (size_t)n * n_cut * 4, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    return NTF_OK;
}
