// Member-skill co-occurrence  C = member^T . skill  on the device (reference src/cmn/team.py:302-337
// `Team.gen_skill_coverage`, which is `scipy.sparse.csr_matrix(np.dot(member.transpose(), skill))` on two uint8 lil matrices with the
// rows of `skipteams` emptied).  Integer / index work, HBM- and latency-bound; nothing here is GEMM shaped:
//
//   1. k_cooc_work     work[m] = sum over the kept teams t containing m of nnz_skill(t)             (atomics on a [M] counter)
//   2. scan            wstart = exclusive scan of work                                               (3-phase block scan, int64)
//   3. k_cooc_expand   seg[wstart[m] ...] = the skill ids of every kept team of m, concatenated      (one thread per (team, member))
//   4a. k_cooc_sort    work[m] <= SORT_MAX: one wave per expert, bitonic sort of the segment in LDS, run-length encode
//   4b. k_cooc_hist    larger rows: one workgroup per expert, u32 histogram over the S skills in an L2-resident scratch row,
//                      ordered compaction of the non-zero bins
//   5. scan of the row lengths -> indptr ; k_cooc_compact copies the rows to their final place
//
// Results are what scipy returns after sort_indices(): counts in uint8 arithmetic (wrap mod 256), entries whose wrapped count is 0
// dropped (scipy's csr_matmat drops zero sums), column ids ascending inside a row.  Bit-exact by construction (integer work).
#include "../../include/opentf_amd.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

namespace {

constexpr int SORT_MAX = 2048;     // entries a single wave sorts in LDS (2 x 8 KiB per workgroup)
constexpr int HIST_WGS = 512;      // persistent workgroups of the histogram path, one scratch row each
constexpr int SCAN_BLOCK = 1024;   // elements per block of the scan (256 threads x 4)

// ---------------------------------------------------------------- scan (exclusive, u32 in -> int64 out, out[n] = total)
__global__ __launch_bounds__(256) void k_scan_block_sums(const uint32_t* __restrict__ in, int64_t n, int64_t* __restrict__ block_sums) {
    __shared__ int64_t red[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
    int64_t s = 0;
    for (int j = 0; j < 4; ++j) { const int64_t i = base + threadIdx.x * 4 + j; if (i < n) s += in[i]; }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_scan_top(int64_t* __restrict__ block_sums, int64_t nblocks) {  // single block, serial over 256-wide strips
    __shared__ int64_t buf[256];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nblocks; b0 += 256) {
        const int64_t i = b0 + threadIdx.x;
        const int64_t v = i < nblocks ? block_sums[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) { const int64_t t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0; __syncthreads(); buf[threadIdx.x] += t; __syncthreads(); }
        if (i < nblocks) block_sums[i] = carry + buf[threadIdx.x] - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry;
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, int64_t n, const int64_t* __restrict__ block_sums,
                                                    int64_t nblocks, int64_t* __restrict__ out) {
    __shared__ int64_t wsum[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x * 4;
    uint32_t v[4]; int64_t s = 0;
    for (int j = 0; j < 4; ++j) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    int64_t inc = s;  // inclusive over the wave
    const int lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { const int64_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int64_t off = block_sums[blockIdx.x] + inc - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += wsum[w];
    for (int j = 0; j < 4; ++j) { if (base + j < n) out[base + j] = off; off += v[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = block_sums[nblocks];
}

// ---------------------------------------------------------------- 1 / 3: per-expert work and the expanded segments
__global__ void k_cooc_work(int64_t n_teams, const int64_t* __restrict__ m_ip, const int32_t* __restrict__ m_ix, const int64_t* __restrict__ s_ip,
                            const uint8_t* __restrict__ skip, uint32_t* __restrict__ work) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_teams || (skip && skip[t])) return;
    const uint32_t ns = (uint32_t)(s_ip[t + 1] - s_ip[t]);
    if (!ns) return;
    for (int64_t p = m_ip[t]; p < m_ip[t + 1]; ++p) atomicAdd(&work[m_ix[p]], ns);
}
__global__ void k_cooc_expand(int64_t n_teams, const int64_t* __restrict__ m_ip, const int32_t* __restrict__ m_ix, const int64_t* __restrict__ s_ip,
                              const int32_t* __restrict__ s_ix, const uint8_t* __restrict__ skip, const int64_t* __restrict__ wstart,
                              uint32_t* __restrict__ cursor, int32_t* __restrict__ seg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_teams || (skip && skip[t])) return;
    const int64_t s0 = s_ip[t];
    const uint32_t ns = (uint32_t)(s_ip[t + 1] - s0);
    if (!ns) return;
    for (int64_t p = m_ip[t]; p < m_ip[t + 1]; ++p) {
        const int m = m_ix[p];
        int32_t* dst = seg + wstart[m] + atomicAdd(&cursor[m], ns);   // the order of a row's teams is irrelevant: the row is counted
        for (uint32_t j = 0; j < ns; ++j) dst[j] = s_ix[s0 + j];
    }
}

// ---------------------------------------------------------------- 4a: short rows, one wave each
__global__ __launch_bounds__(64) void k_cooc_sort(int n_experts, const uint32_t* __restrict__ work, const int64_t* __restrict__ wstart,
                                                  const int32_t* __restrict__ seg, int32_t* __restrict__ tmp_ix, uint8_t* __restrict__ tmp_val,
                                                  uint32_t* __restrict__ row_len, int32_t* __restrict__ big_list, uint32_t* __restrict__ big_count) {
    __shared__ uint32_t a[SORT_MAX];
    __shared__ uint32_t head[SORT_MAX];
    const int m = blockIdx.x, lane = threadIdx.x;
    const uint32_t w = work[m];
    if (w == 0) { if (lane == 0) row_len[m] = 0; return; }
    if (w > (uint32_t)SORT_MAX) { if (lane == 0) big_list[atomicAdd(big_count, 1u)] = m; return; }
    const int64_t base = wstart[m];
    int n = 64; while (n < (int)w) n <<= 1;
    for (int i = lane; i < n; i += 64) a[i] = i < (int)w ? (uint32_t)seg[base + i] : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= n; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < n; i += 64) {
                const int q = i ^ j;
                if (q > i) {
                    const uint32_t x = a[i], y = a[q];
                    if ((x > y) == ((i & k) == 0)) { a[i] = y; a[q] = x; }
                }
            }
            __syncthreads();
        }
    // run heads, in order
    int nheads = 0;
    for (int i0 = 0; i0 < (int)w; i0 += 64) {
        const int i = i0 + lane;
        const bool h = i < (int)w && (i == 0 || a[i] != a[i - 1]);
        const uint64_t mask = __ballot(h);
        if (h) head[nheads + __popcll(mask & ((1ull << lane) - 1))] = (uint32_t)i;
        nheads += __popcll(mask);
    }
    __syncthreads();
    int nout = 0;
    for (int r0 = 0; r0 < nheads; r0 += 64) {
        const int r = r0 + lane;
        uint32_t pos = 0, cnt = 0;
        if (r < nheads) { pos = head[r]; cnt = ((r + 1 < nheads ? head[r + 1] : w) - pos) & 255u; }   // uint8 arithmetic
        const bool keep = cnt != 0;                                                                  // zero sums are not stored
        const uint64_t mask = __ballot(keep);
        if (keep) { const int o = nout + __popcll(mask & ((1ull << lane) - 1)); tmp_ix[base + o] = (int32_t)a[pos]; tmp_val[base + o] = (uint8_t)cnt; }
        nout += __popcll(mask);
    }
    if (lane == 0) row_len[m] = (uint32_t)nout;
}

// ---------------------------------------------------------------- 4b: long rows, histogram over the skills
__global__ __launch_bounds__(256) void k_cooc_hist(const int32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count, int n_skills,
                                                   const uint32_t* __restrict__ work, const int64_t* __restrict__ wstart,
                                                   const int32_t* __restrict__ seg, uint32_t* __restrict__ scratch /*[grid, n_skills] zero*/,
                                                   int32_t* __restrict__ tmp_ix, uint8_t* __restrict__ tmp_val, uint32_t* __restrict__ row_len) {
    __shared__ int wtot[4];
    __shared__ int run;
    uint32_t* hist = scratch + (size_t)blockIdx.x * n_skills;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t nbig = *big_count;
    for (uint32_t bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const int m = big_list[bi];
        const int64_t base = wstart[m];
        const uint32_t w = work[m];
        for (uint32_t i = threadIdx.x; i < w; i += 256) atomicAdd(&hist[seg[base + i]], 1u);
        if (threadIdx.x == 0) run = 0;
        __threadfence();
        __syncthreads();
        for (int b0 = 0; b0 < n_skills; b0 += 1024) {
            const int b = b0 + threadIdx.x * 4;
            uint32_t c[4]; int keep = 0;
            for (int j = 0; j < 4; ++j) {
                c[j] = b + j < n_skills ? __hip_atomic_load(&hist[b + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // L2, not a stale L1 line
                if (c[j]) hist[b + j] = 0;
                c[j] &= 255u;
                keep += c[j] != 0;
            }
            int inc = keep;
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
            if (lane == 63) wtot[wv] = inc;
            __syncthreads();
            int off = run + inc - keep;
            for (int q = 0; q < wv; ++q) off += wtot[q];
            for (int j = 0; j < 4; ++j) if (c[j]) { tmp_ix[base + off] = b + j; tmp_val[base + off] = (uint8_t)c[j]; ++off; }
            __syncthreads();
            if (threadIdx.x == 255) run = off;   // the last thread's running offset = everything kept so far
            __syncthreads();
        }
        if (threadIdx.x == 0) row_len[m] = (uint32_t)run;
        __threadfence();
        __syncthreads();
    }
}

// ---------------------------------------------------------------- 5: rows to their final place
__global__ __launch_bounds__(256) void k_cooc_compact(int n_experts, const uint32_t* __restrict__ row_len, const int64_t* __restrict__ wstart,
                                                      const int64_t* __restrict__ indptr, const int32_t* __restrict__ tmp_ix,
                                                      const uint8_t* __restrict__ tmp_val, int32_t* __restrict__ out_ix, uint8_t* __restrict__ out_val) {
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (m >= n_experts) return;
    const uint32_t n = row_len[m];
    const int64_t src = wstart[m], dst = indptr[m];
    for (uint32_t i = lane; i < n; i += 64) { out_ix[dst + i] = tmp_ix[src + i]; out_val[dst + i] = tmp_val[src + i]; }
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) hipFree(p); }
    bool alloc(size_t bytes, bool zero = false) {
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { p = nullptr; return false; }
        return !zero || hipMemset(p, 0, bytes ? bytes : 16) == hipSuccess;
    }
    bool put(const void* host, size_t bytes) { return alloc(bytes) && (!bytes || hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) == hipSuccess); }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

bool scan_u32(const uint32_t* in, int64_t n, int64_t* out /*[n+1]*/) {
    const int64_t nb = std::max<int64_t>(1, (n + SCAN_BLOCK - 1) / SCAN_BLOCK);
    Buf sums;
    if (!sums.alloc((size_t)(nb + 1) * 8)) return false;
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(256), 0, 0, in, n, sums.as<int64_t>());
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, 0, sums.as<int64_t>(), nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, 0, in, n, sums.as<int64_t>(), nb, out);
    return hipDeviceSynchronize() == hipSuccess;
}

}  // namespace

struct ntf_csr_result {
    int device = 0;
    int64_t n_rows = 0, nnz = 0;
    int64_t* indptr = nullptr; int32_t* indices = nullptr; uint8_t* data = nullptr;   // device
    double ms = 0;                                                                       // device time of the build (HIP events)
};

extern "C" int ntf_skill_cooccurrence(int device, int64_t n_teams, int32_t n_members, int32_t n_skills, const int64_t* m_indptr,
                                      const int32_t* m_indices, const int64_t* s_indptr, const int32_t* s_indices, const int64_t* skip_rows,
                                      int64_t n_skip, ntf_csr_result** out, int64_t* nnz_out) {
    if (!out) return NTF_EINVAL;
    *out = nullptr;
    if (n_teams < 1 || n_members < 1 || n_skills < 1 || !m_indptr || !s_indptr || n_skip < 0 || (n_skip && !skip_rows)) return NTF_EINVAL;
    const int64_t m_nnz = m_indptr[n_teams], s_nnz = s_indptr[n_teams];
    if ((m_nnz && !m_indices) || (s_nnz && !s_indices)) return NTF_EINVAL;
    for (int64_t p = 0; p < m_nnz; ++p) if (m_indices[p] < 0 || m_indices[p] >= n_members) return NTF_EINVAL;
    for (int64_t p = 0; p < s_nnz; ++p) if (s_indices[p] < 0 || s_indices[p] >= n_skills) return NTF_EINVAL;
    std::vector<uint8_t> skip;
    if (n_skip) {
        skip.assign((size_t)n_teams, 0);
        for (int64_t i = 0; i < n_skip; ++i) { if (skip_rows[i] < 0 || skip_rows[i] >= n_teams) return NTF_EINVAL; skip[(size_t)skip_rows[i]] = 1; }
    }
    if (hipSetDevice(device) != hipSuccess) return NTF_EHIP;
    Buf mip, mix, sip, six, dskip, work, cursor, wstart, seg, tix, tval, rlen, biglist, bigcount, scratch;
    if (!mip.put(m_indptr, (size_t)(n_teams + 1) * 8) || !mix.put(m_indices, (size_t)m_nnz * 4) || !sip.put(s_indptr, (size_t)(n_teams + 1) * 8) ||
        !six.put(s_indices, (size_t)s_nnz * 4) || (n_skip && !dskip.put(skip.data(), (size_t)n_teams)) || !work.alloc((size_t)n_members * 4, true) ||
        !cursor.alloc((size_t)n_members * 4, true) || !wstart.alloc((size_t)(n_members + 1) * 8) || !rlen.alloc((size_t)n_members * 4, true) ||
        !biglist.alloc((size_t)n_members * 4) || !bigcount.alloc(4, true)) return NTF_ENOMEM;
    hipEvent_t ev0, ev1;
    hipEventCreate(&ev0); hipEventCreate(&ev1);
    hipEventRecord(ev0, 0);
    const unsigned tb = (unsigned)((n_teams + 255) / 256);
    hipLaunchKernelGGL(k_cooc_work, dim3(tb), dim3(256), 0, 0, n_teams, mip.as<int64_t>(), mix.as<int32_t>(), sip.as<int64_t>(),
                       n_skip ? dskip.as<uint8_t>() : nullptr, work.as<uint32_t>());
    if (!scan_u32(work.as<uint32_t>(), n_members, wstart.as<int64_t>())) return NTF_EHIP;
    int64_t total = 0;
    if (hipMemcpy(&total, wstart.as<int64_t>() + n_members, 8, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    if (!seg.alloc((size_t)total * 4) || !tix.alloc((size_t)total * 4) || !tval.alloc((size_t)total)) return NTF_ENOMEM;
    hipLaunchKernelGGL(k_cooc_expand, dim3(tb), dim3(256), 0, 0, n_teams, mip.as<int64_t>(), mix.as<int32_t>(), sip.as<int64_t>(), six.as<int32_t>(),
                       n_skip ? dskip.as<uint8_t>() : nullptr, wstart.as<int64_t>(), cursor.as<uint32_t>(), seg.as<int32_t>());
    hipLaunchKernelGGL(k_cooc_sort, dim3((unsigned)n_members), dim3(64), 0, 0, n_members, work.as<uint32_t>(), wstart.as<int64_t>(), seg.as<int32_t>(),
                       tix.as<int32_t>(), tval.as<uint8_t>(), rlen.as<uint32_t>(), biglist.as<int32_t>(), bigcount.as<uint32_t>());
    uint32_t nbig = 0;
    if (hipMemcpy(&nbig, bigcount.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    if (nbig) {
        const int wgs = (int)std::min<uint32_t>(nbig, HIST_WGS);
        if (!scratch.alloc((size_t)wgs * n_skills * 4, true)) return NTF_ENOMEM;
        hipLaunchKernelGGL(k_cooc_hist, dim3(wgs), dim3(256), 0, 0, biglist.as<int32_t>(), bigcount.as<uint32_t>(), n_skills, work.as<uint32_t>(),
                           wstart.as<int64_t>(), seg.as<int32_t>(), scratch.as<uint32_t>(), tix.as<int32_t>(), tval.as<uint8_t>(), rlen.as<uint32_t>());
    }
    ntf_csr_result* r = new ntf_csr_result();
    r->device = device; r->n_rows = n_members;
    if (hipMalloc(&r->indptr, (size_t)(n_members + 1) * 8) != hipSuccess) { delete r; return NTF_ENOMEM; }
    if (!scan_u32(rlen.as<uint32_t>(), n_members, r->indptr) ||
        hipMemcpy(&r->nnz, r->indptr + n_members, 8, hipMemcpyDeviceToHost) != hipSuccess) { hipFree(r->indptr); delete r; return NTF_EHIP; }
    if (hipMalloc(&r->indices, (size_t)std::max<int64_t>(r->nnz, 4) * 4) != hipSuccess || hipMalloc(&r->data, (size_t)std::max<int64_t>(r->nnz, 16)) != hipSuccess) {
        hipFree(r->indptr); if (r->indices) hipFree(r->indices); delete r; return NTF_ENOMEM;
    }
    hipLaunchKernelGGL(k_cooc_compact, dim3((unsigned)((n_members + 3) / 4)), dim3(256), 0, 0, n_members, rlen.as<uint32_t>(), wstart.as<int64_t>(),
                       r->indptr, tix.as<int32_t>(), tval.as<uint8_t>(), r->indices, r->data);
    hipEventRecord(ev1, 0);
    const hipError_t s = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, ev0, ev1); r->ms = ms;
    hipEventDestroy(ev0); hipEventDestroy(ev1);
    if (s != hipSuccess || hipGetLastError() != hipSuccess) { ntf_csr_result_free(r); return NTF_EHIP; }
    if (nnz_out) *nnz_out = r->nnz;
    *out = r;
    return NTF_OK;
}

extern "C" int ntf_csr_result_fetch(ntf_csr_result* r, int64_t* indptr, int32_t* indices, uint8_t* data, double* device_ms) {
    if (!r || !indptr) return NTF_EINVAL;
    if (hipSetDevice(r->device) != hipSuccess) return NTF_EHIP;
    if (hipMemcpy(indptr, r->indptr, (size_t)(r->n_rows + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    if (r->nnz && indices && hipMemcpy(indices, r->indices, (size_t)r->nnz * 4, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    if (r->nnz && data && hipMemcpy(data, r->data, (size_t)r->nnz, hipMemcpyDeviceToHost) != hipSuccess) return NTF_EHIP;
    if (device_ms) *device_ms = r->ms;
    return NTF_OK;
}

extern "C" void ntf_csr_result_free(ntf_csr_result* r) {
    if (!r) return;
    hipSetDevice(r->device);
    if (r->indptr) hipFree(r->indptr);
    if (r->indices) hipFree(r->indices);
    if (r->data) hipFree(r->data);
    delete r;
}
