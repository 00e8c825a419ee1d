// node2vec skill-embedding producer on the MI355X: the table E [S, d] that the team2vec gather (k_gather_pool) consumes.
// Replaces the node2vec branch of the reference, src/mdl/emb/gnn.py:153-168 (torch_geometric.nn.Node2Vec, pinned 2.6.1 in requirements.txt:51,
// not installed here: its published algorithm is restated in oracle/n2v_oracle.py) and its training loop _train_rw, gnn.py:401-453:
//   loader  : per batch of start nodes, `walks_per_node` uniform random walks (p = q = 1) of `walk_length` nodes, cut into windows of
//             `context_size`; negatives = the same start nodes followed by uniformly random nodes
//   loss    : -mean log(sigmoid(<e_start, e_rest>) + 1e-15) over positive pairs  - mean log(1 - sigmoid(.) + 1e-15) over negative pairs
//   update  : torch.optim.Adam on the dense embedding matrix (every row moves every step through its moments)
// One wave per window row: the start row lives in registers (d / 64 values per lane), every rest row costs one coalesced row read, a wave
// reduction and one coalesced f32 atomic row add (the forms that run at full rate on gfx950: 256 contiguous bytes per wave-instruction).
#include "../../include/opentf_amd.h"
#include "ntf_device.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

using namespace ntf;

struct ntf_n2v {
    int device = 0; hipStream_t st = nullptr;
    int64_t n = 0; int d = 0, d_user = 0; int64_t nnz = 0;      // d: the device row stride (a multiple of 64); d_user: the embedding size the host sees
    int64_t* rowptr = nullptr; int32_t* col = nullptr;
    float *W = nullptr, *G = nullptr, *M1 = nullptr, *V2 = nullptr;
    int64_t* d_batch = nullptr; int64_t batch_cap = 0;
    int64_t* d_rows = nullptr; int64_t rows_cap = 0;     // window rows [n_rows, ctx]
    double* d_loss = nullptr;
    int64_t adam_t = 0; uint64_t seed = 0, step = 0;
    std::string err;
};
static thread_local std::string g_n2v_create_error;

#define NCHK(h, call) do { hipError_t _s = (call); if (_s != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(_s); return NTF_EHIP; } } while (0)
#define NFAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

extern "C" const char* ntf_n2v_last_error(const ntf_n2v* h) { return h ? h->err.c_str() : g_n2v_create_error.c_str(); }

// uniform random walk (torch_cluster / pyg-lib semantics for p = q = 1: a node without neighbours keeps the walk where it is).
// Row r starts at batch[r % B] (`batch.repeat(walks_per_node)`); one thread per walk, one Philox call per 4 steps.
__global__ void k_n2v_walks(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int64_t* __restrict__ batch, int64_t B,
                            int64_t n_walks, int wl, uint32_t k0, uint32_t k1, uint32_t step, int64_t* __restrict__ rw) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_walks) return;
    int64_t cur = batch[r % B];
    rw[r * wl] = cur;
    uint4 rnd = make_uint4(0, 0, 0, 0);
    for (int s = 1; s < wl; ++s) {
        if (((s - 1) & 3) == 0) rnd = philox4x32(make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)((s - 1) >> 2), step), make_uint2(k0, k1));
        const uint32_t u = ((s - 1) & 3) == 0 ? rnd.x : ((s - 1) & 3) == 1 ? rnd.y : ((s - 1) & 3) == 2 ? rnd.z : rnd.w;
        const int64_t a = rowptr[cur], deg = rowptr[cur + 1] - a;
        if (deg > 0) cur = col[a + (int64_t)(((uint64_t)u * (uint64_t)deg) >> 32)];
        rw[r * wl + s] = cur;
    }
}
// negative rows: start node followed by uniformly random nodes (`torch.randint(num_nodes, ...)`), start = batch[r % B]
__global__ void k_n2v_negs(const int64_t* __restrict__ batch, int64_t B, int64_t n_rows, int wl, int64_t num_nodes, uint32_t k0, uint32_t k1, uint32_t step,
                           int64_t* __restrict__ rw) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    rw[r * wl] = batch[r % B];
    uint4 rnd = make_uint4(0, 0, 0, 0);
    for (int s = 1; s < wl; ++s) {
        if (((s - 1) & 3) == 0) rnd = philox4x32(make_uint4((uint32_t)r, (uint32_t)(r >> 32), 0x4E454700u + (uint32_t)((s - 1) >> 2), step), make_uint2(k0, k1));
        const uint32_t u = ((s - 1) & 3) == 0 ? rnd.x : ((s - 1) & 3) == 1 ? rnd.y : ((s - 1) & 3) == 2 ? rnd.z : rnd.w;
        rw[r * wl + s] = (int64_t)(((uint64_t)u * (uint64_t)num_nodes) >> 32);
    }
}
// windows of `ctx` consecutive nodes: out row j * n_walks + r = rw[r, j : j + ctx]   (torch.cat of the slices along dim 0)
__global__ void k_n2v_windows(const int64_t* __restrict__ rw, int64_t n_walks, int wl, int ctx, int64_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = wl + 1 - ctx;
    if (t >= n_walks * nw * ctx) return;
    const int c = (int)(t % ctx); const int64_t row = t / ctx, j = row / n_walks, r = row % n_walks;
    out[t] = rw[r * wl + j + c];
}

// loss and gradient of one set of window rows; sign = +1 positive pairs, -1 negative pairs.  NV = d / 64 values per lane.
template <int NV>
__global__ __launch_bounds__(256) void k_n2v_pairs(const float* __restrict__ W, const int64_t* __restrict__ rows, int64_t n_rows, int ctx, int d,
                                                   float inv_pairs, int positive, float* __restrict__ G, double* __restrict__ loss) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float lsum = 0.f;
    if (r < n_rows) {
        const int64_t s = rows[r * ctx];
        float hs[NV], gs[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) { hs[k] = W[s * d + lane + 64 * k]; gs[k] = 0.f; }
        for (int c = 1; c < ctx; ++c) {
            const int64_t v = rows[r * ctx + c];
            float hv[NV], dot = 0.f;
#pragma unroll
            for (int k = 0; k < NV; ++k) { hv[k] = W[v * d + lane + 64 * k]; dot += hs[k] * hv[k]; }
            dot = wave_reduce_sum(dot);
            const float sg = 1.f / (1.f + expf(-dot));
            float coef;   // d loss / d dot
            if (positive) { lsum -= logf(sg + 1e-15f); coef = -sg * (1.f - sg) / (sg + 1e-15f) * inv_pairs; }
            else { lsum -= logf(1.f - sg + 1e-15f); coef = sg * (1.f - sg) / (1.f - sg + 1e-15f) * inv_pairs; }
#pragma unroll
            for (int k = 0; k < NV; ++k) { atomicAdd(G + v * d + lane + 64 * k, coef * hs[k]); gs[k] += coef * hv[k]; }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) atomicAdd(G + s * d + lane + 64 * k, gs[k]);
    }
    if (lane == 0 && r < n_rows) atomicAdd(loss, (double)lsum * (double)inv_pairs);
}

// dense Adam that also clears the gradient for the next batch (one pass over the four buffers)
__global__ void k_n2v_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_over_bc1,
                           float b1, float b2, float eps, float bc2_sqrt) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < (n >> 2); q += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[q], mm = reinterpret_cast<float4*>(m)[q], vv = reinterpret_cast<float4*>(v)[q];
        const float4 gg = reinterpret_cast<float4*>(g)[q];
        float* P = &pp.x; float* M = &mm.x; float* V = &vv.x; const float* Gv = &gg.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            M[k] = M[k] * b1 + (1.f - b1) * Gv[k];
            V[k] = V[k] * b2 + (1.f - b2) * Gv[k] * Gv[k];
            P[k] -= lr_over_bc1 * (M[k] / (sqrtf(V[k]) / bc2_sqrt + eps));
        }
        reinterpret_cast<float4*>(p)[q] = pp; reinterpret_cast<float4*>(m)[q] = mm; reinterpret_cast<float4*>(v)[q] = vv;
        reinterpret_cast<float4*>(g)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// v_loss of _train_rw (gnn.py:420-431): mean BCE-with-logits of the held-out edges' scores against 1 = mean softplus(-score)
__global__ __launch_bounds__(256) void k_n2v_edge_bce(const float* __restrict__ W, const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t n, int d,
                                                      double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float dot = 0.f;
    for (int k = lane; k < d; k += 64) dot += W[src[r] * d + k] * W[dst[r] * d + k];
    dot = wave_reduce_sum(dot);
    if (lane == 0) atomicAdd(out, (double)(fmaxf(-dot, 0.f) + log1pf(expf(-fabsf(dot)))));
}

template <typename T> static int nalloc(ntf_n2v* h, T** p, int64_t n) {
    *p = nullptr;
    if (n <= 0) return NTF_OK;
    if (hipMalloc((void**)p, (size_t)n * sizeof(T)) != hipSuccess) { h->err = "hipMalloc failed"; return NTF_ENOMEM; }
    return NTF_OK;
}

extern "C" void ntf_n2v_destroy(ntf_n2v* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->st) hipStreamSynchronize(h->st);
    for (void* p : {(void*)h->rowptr, (void*)h->col, (void*)h->W, (void*)h->G, (void*)h->M1, (void*)h->V2, (void*)h->d_batch, (void*)h->d_rows, (void*)h->d_loss})
        if (p) hipFree(p);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

extern "C" int ntf_n2v_create(int device, int64_t num_nodes, int32_t d, const int64_t* rowptr, const int32_t* col, const float* init_weight, uint64_t seed, ntf_n2v** out) {
    if (!out) { g_n2v_create_error = "out is NULL"; return NTF_EINVAL; }
    *out = nullptr;
    if (num_nodes < 1 || d < 1 || d > 256 || !rowptr || !init_weight) { g_n2v_create_error = "n2v: need num_nodes >= 1, 1 <= d <= 256, a CSR graph and initial weights"; return NTF_EINVAL; }
    const int64_t nnz = rowptr[num_nodes];
    for (int64_t i = 0; i < num_nodes; ++i) if (rowptr[i + 1] < rowptr[i]) { g_n2v_create_error = "n2v: rowptr not monotone"; return NTF_EINVAL; }
    for (int64_t p = 0; p < nnz; ++p) if (col[p] < 0 || col[p] >= num_nodes) { g_n2v_create_error = "n2v: neighbour id out of range"; return NTF_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_n2v_create_error = "no such HIP device (there is no CPU fallback)"; return NTF_EHIP; }
    hipSetDevice(device);
    ntf_n2v* h = new ntf_n2v();
    // any embedding size 1..256 (round 6; data.embedding.d is free in the reference): a device row is padded to a multiple of 64 floats (h->d, what the kernels see); the pad
    // columns of the table are zero and stay zero - their gradient is a multiple of another row's zero pad, and Adam leaves a zero parameter with zero moments where it is
    h->device = device; h->n = num_nodes; h->d_user = d; h->d = (d + 63) / 64 * 64; h->nnz = nnz; h->seed = seed;
    const int du = d; d = h->d;
    int rc = NTF_OK;
    auto A = [&](int r) { if (rc == NTF_OK) rc = r; };
    if (hipStreamCreate(&h->st) != hipSuccess) { g_n2v_create_error = "hipStreamCreate failed"; delete h; return NTF_EHIP; }
    const int64_t np = num_nodes * d;
    A(nalloc(h, &h->rowptr, num_nodes + 1)); A(nalloc(h, &h->col, std::max<int64_t>(nnz, 1)));
    A(nalloc(h, &h->W, np)); A(nalloc(h, &h->G, np)); A(nalloc(h, &h->M1, np)); A(nalloc(h, &h->V2, np)); A(nalloc(h, &h->d_loss, 2));
    if (rc != NTF_OK) { g_n2v_create_error = h->err; ntf_n2v_destroy(h); return rc; }
    hipMemcpy(h->rowptr, rowptr, (num_nodes + 1) * 8, hipMemcpyHostToDevice);
    if (nnz) hipMemcpy(h->col, col, nnz * 4, hipMemcpyHostToDevice);
    if (du != d) hipMemset(h->W, 0, np * 4);
    hipMemcpy2D(h->W, (size_t)d * 4, init_weight, (size_t)du * 4, (size_t)du * 4, (size_t)num_nodes, hipMemcpyHostToDevice);
    hipMemsetAsync(h->G, 0, np * 4, h->st); hipMemsetAsync(h->M1, 0, np * 4, h->st); hipMemsetAsync(h->V2, 0, np * 4, h->st);
    if (hipStreamSynchronize(h->st) != hipSuccess) { g_n2v_create_error = "device initialisation failed"; ntf_n2v_destroy(h); return NTF_EHIP; }
    *out = h;
    return NTF_OK;
}

static void n2v_key(const ntf_n2v* h, uint64_t step, int tensor, uint32_t& k0, uint32_t& k1) {
    uint64_t x = h->seed ^ (step * 0x9E3779B97F4A7C15ull + (uint64_t)tensor * 0xBF58476D1CE4E5B9ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    k0 = (uint32_t)x; k1 = (uint32_t)(x >> 32);
}
static int stage_batch(ntf_n2v* h, const int64_t* batch, int64_t B) {
    for (int64_t i = 0; i < B; ++i) if (batch[i] < 0 || batch[i] >= h->n) NFAIL(h, NTF_EINVAL, "n2v: start node out of range");
    if (h->batch_cap < B) { if (h->d_batch) hipFree(h->d_batch); int r = nalloc(h, &h->d_batch, B); if (r) return r; h->batch_cap = B; }
    NCHK(h, hipMemcpyAsync(h->d_batch, batch, B * 8, hipMemcpyHostToDevice, h->st));
    NCHK(h, hipStreamSynchronize(h->st));
    return NTF_OK;
}
static int need_rows(ntf_n2v* h, int64_t elems) {
    if (h->rows_cap < elems) { if (h->d_rows) hipFree(h->d_rows); int r = nalloc(h, &h->d_rows, elems); if (r) return r; h->rows_cap = elems; }
    return NTF_OK;
}

extern "C" int ntf_n2v_walks(ntf_n2v* h, const int64_t* start, int64_t n, int32_t walk_length, uint64_t step, int64_t* out_host) {
    if (!h || !start || n < 1 || walk_length < 1 || !out_host) return NTF_EINVAL;
    NCHK(h, hipSetDevice(h->device));
    int r = stage_batch(h, start, n); if (r) return r;
    if ((r = need_rows(h, n * walk_length))) return r;
    uint32_t k0, k1; n2v_key(h, step, 0, k0, k1);
    hipLaunchKernelGGL(k_n2v_walks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->st, h->rowptr, h->col, h->d_batch, n, n, walk_length, k0, k1, (uint32_t)step, h->d_rows);
    NCHK(h, hipMemcpyAsync(out_host, h->d_rows, n * walk_length * 8, hipMemcpyDeviceToHost, h->st));
    NCHK(h, hipStreamSynchronize(h->st));
    return NTF_OK;
}

static void launch_pairs(ntf_n2v* h, const int64_t* rows, int64_t n_rows, int ctx, int positive) {
    if (n_rows <= 0 || ctx < 2) return;
    const float inv = 1.f / (float)(n_rows * (ctx - 1));
    const dim3 grid((unsigned)((n_rows + 3) / 4)), block(256);
    switch (h->d / 64) {
        case 1: hipLaunchKernelGGL(k_n2v_pairs<1>, grid, block, 0, h->st, h->W, rows, n_rows, ctx, h->d, inv, positive, h->G, h->d_loss); break;
        case 2: hipLaunchKernelGGL(k_n2v_pairs<2>, grid, block, 0, h->st, h->W, rows, n_rows, ctx, h->d, inv, positive, h->G, h->d_loss); break;
        case 3: hipLaunchKernelGGL(k_n2v_pairs<3>, grid, block, 0, h->st, h->W, rows, n_rows, ctx, h->d, inv, positive, h->G, h->d_loss); break;
        default: hipLaunchKernelGGL(k_n2v_pairs<4>, grid, block, 0, h->st, h->W, rows, n_rows, ctx, h->d, inv, positive, h->G, h->d_loss); break;
    }
}

extern "C" int ntf_n2v_train_batch(ntf_n2v* h, const int64_t* batch, int32_t B, int32_t walk_length, int32_t context, int32_t walks_per_node, int32_t num_neg,
                                   float lr, const int64_t* inj_pos, int64_t n_pos, const int64_t* inj_neg, int64_t n_neg, int32_t apply, float* loss_out) {
    if (!h || walk_length < 2 || context < 2 || context > walk_length) return NTF_EINVAL;
    NCHK(h, hipSetDevice(h->device));
    NCHK(h, hipMemsetAsync(h->d_loss, 0, 8, h->st));
    const uint64_t step = h->step++;
    int r;
    if (inj_pos || inj_neg) {   // parity tests: the window rows themselves are given
        if ((r = need_rows(h, (n_pos + n_neg) * context))) return r;
        if ((n_pos > 0 && !inj_pos) || (n_neg > 0 && !inj_neg) || n_pos < 0 || n_neg < 0) return NTF_EINVAL;
        for (int64_t i = 0; i < n_pos * context; ++i) if (inj_pos[i] < 0 || inj_pos[i] >= h->n) NFAIL(h, NTF_EINVAL, "n2v: injected node out of range");
        for (int64_t i = 0; i < n_neg * context; ++i) if (inj_neg[i] < 0 || inj_neg[i] >= h->n) NFAIL(h, NTF_EINVAL, "n2v: injected node out of range");
        if (n_pos) NCHK(h, hipMemcpyAsync(h->d_rows, inj_pos, n_pos * context * 8, hipMemcpyHostToDevice, h->st));
        if (n_neg) NCHK(h, hipMemcpyAsync(h->d_rows + n_pos * context, inj_neg, n_neg * context * 8, hipMemcpyHostToDevice, h->st));
        NCHK(h, hipStreamSynchronize(h->st));
        launch_pairs(h, h->d_rows, n_pos, context, 1);
        launch_pairs(h, h->d_rows + n_pos * context, n_neg, context, 0);
    } else {
        if (!batch || B < 1 || walks_per_node < 1 || num_neg < 0) return NTF_EINVAL;
        if ((r = stage_batch(h, batch, B))) return r;
        const int nw = walk_length + 1 - context;
        const int64_t pw = (int64_t)B * walks_per_node, ng = pw * num_neg;
        // scratch: [pos walks | neg walks | pos windows | neg windows]
        if ((r = need_rows(h, (pw + ng) * walk_length + (pw + ng) * nw * context))) return r;
        int64_t* rw_p = h->d_rows; int64_t* rw_n = rw_p + pw * walk_length;
        int64_t* win_p = rw_n + ng * walk_length; int64_t* win_n = win_p + pw * nw * context;
        uint32_t k0, k1;
        n2v_key(h, step, 0, k0, k1);
        hipLaunchKernelGGL(k_n2v_walks, dim3((unsigned)((pw + 255) / 256)), dim3(256), 0, h->st, h->rowptr, h->col, h->d_batch, (int64_t)B, pw, walk_length, k0, k1, (uint32_t)step, rw_p);
        hipLaunchKernelGGL(k_n2v_windows, dim3((unsigned)((pw * nw * context + 255) / 256)), dim3(256), 0, h->st, rw_p, pw, walk_length, context, win_p);
        launch_pairs(h, win_p, pw * nw, context, 1);
        if (ng) {
            n2v_key(h, step, 1, k0, k1);
            hipLaunchKernelGGL(k_n2v_negs, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, h->st, h->d_batch, (int64_t)B, ng, walk_length, h->n, k0, k1, (uint32_t)step, rw_n);
            hipLaunchKernelGGL(k_n2v_windows, dim3((unsigned)((ng * nw * context + 255) / 256)), dim3(256), 0, h->st, rw_n, ng, walk_length, context, win_n);
            launch_pairs(h, win_n, ng * nw, context, 0);
        }
    }
    if (apply) {
        h->adam_t += 1;
        const double b1 = 0.9, b2 = 0.999;
        const double bc1 = 1.0 - std::pow(b1, (double)h->adam_t), bc2 = 1.0 - std::pow(b2, (double)h->adam_t);
        const int64_t np = h->n * h->d;
        const int blocks = (int)std::min<int64_t>((np / 4 + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(k_n2v_adam, dim3(blocks), dim3(256), 0, h->st, h->W, h->G, h->M1, h->V2, np, lr / (float)bc1, (float)b1, (float)b2, 1e-8f, (float)std::sqrt(bc2));
    }
    if (loss_out) {
        double l = 0;
        NCHK(h, hipMemcpyAsync(&l, h->d_loss, 8, hipMemcpyDeviceToHost, h->st));
        NCHK(h, hipStreamSynchronize(h->st));
        *loss_out = (float)l;
    }
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) NFAIL(h, NTF_EHIP, std::string("n2v kernel launch: ") + hipGetErrorString(s));
    return NTF_OK;
}

extern "C" int ntf_n2v_get(ntf_n2v* h, int what, float* host) {   // what: 0 = embedding.weight, 1 = its gradient (as the last batch with apply = 0 left it)
    if (!h || !host || what < 0 || what > 1) return NTF_EINVAL;
    NCHK(h, hipSetDevice(h->device));
    NCHK(h, hipStreamSynchronize(h->st));
    NCHK(h, hipMemcpy2D(host, (size_t)h->d_user * 4, what ? h->G : h->W, (size_t)h->d * 4, (size_t)h->d_user * 4, (size_t)h->n, hipMemcpyDeviceToHost));
    if (what) NCHK(h, hipMemsetAsync(h->G, 0, (size_t)h->n * h->d * 4, h->st));   // reading the gradient consumes it
    return NTF_OK;
}

extern "C" int ntf_n2v_edge_bce(ntf_n2v* h, const int64_t* src, const int64_t* dst, int64_t n, float* mean_bce) {
    if (!h || !src || !dst || n < 1 || !mean_bce) return NTF_EINVAL;
    NCHK(h, hipSetDevice(h->device));
    for (int64_t i = 0; i < n; ++i) if (src[i] < 0 || src[i] >= h->n || dst[i] < 0 || dst[i] >= h->n) NFAIL(h, NTF_EINVAL, "n2v: edge endpoint out of range");
    int r = need_rows(h, 2 * n); if (r) return r;
    NCHK(h, hipMemcpyAsync(h->d_rows, src, n * 8, hipMemcpyHostToDevice, h->st));
    NCHK(h, hipMemcpyAsync(h->d_rows + n, dst, n * 8, hipMemcpyHostToDevice, h->st));
    NCHK(h, hipMemsetAsync(h->d_loss + 1, 0, 8, h->st));
    hipLaunchKernelGGL(k_n2v_edge_bce, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, h->st, h->W, h->d_rows, h->d_rows + n, n, h->d, h->d_loss + 1);
    double l = 0;
    NCHK(h, hipMemcpyAsync(&l, h->d_loss + 1, 8, hipMemcpyDeviceToHost, h->st));
    NCHK(h, hipStreamSynchronize(h->st));
    *mean_bce = (float)(l / (double)n);
    return NTF_OK;
}
