// doc2vec team-vector producer on the MI355X: the table [n_teams, d] that D2v.get_dense_vecs hands to the Fnn / Bnn input.
// Replaces gensim.models.Doc2Vec as the reference calls it, src/mdl/emb/d2v.py:69-84 (gensim==4.3.3, requirements.txt:44, not installed here: its published
// algorithm - word2vec.py prepare_vocab / make_cum_table, doc2vec_inner.pyx fast_document_dm_neg / fast_document_dbow_neg - is restated in
// oracle/d2v_oracle.py, which also says what is pinned against the gensim objects the reference's authors committed):
//   PV-DM   (dm = 1, cbow_mean = 1)  per kept position i of a document: l1 = mean(doc vector, word vectors of the shrunk window); the word and `negative`
//           table draws: f = l1 . syn1neg[t], skipped when |f| >= 6, g = (label - sigmoid_table(f)) * alpha, work += g syn1neg[t], syn1neg[t] += g l1;
//           doc vector and window word vectors += work
//   PV-DBOW (dm = 0, dbow_words = 1) per kept position i: the same unit with every window word vector as the input (input += work), then with the doc vector
// gensim trains Hogwild on all cores (the reference's log of dblp mt10.ts2: 224 workers, 276 s per epoch of 19.07 M words, 100 epochs = 7.7 h).  Here: one
// wave per document, the doc vector in registers for the whole document (d / 64 values per lane), word / syn1neg rows as one coalesced 512-B row read and
// one coalesced f32 atomic row add each, every random draw a Philox word counted by (document, position, unit, slot) - so the result does not depend on how
// the documents are spread over waves except through the order of the atomic adds, and a one-wave launch (`serial`) reproduces the oracle's sequential pass.
#include "../../include/opentf_amd.h"
#include "ntf_device.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

using namespace ntf;

struct ntf_d2v {
    int device = 0; hipStream_t st = nullptr;
    int64_t n_docs = 0, n_vocab = 0, n_words = 0; int d = 0, dp = 0;   // dp: the device row stride, d rounded up to a multiple of 64 (a wave holds dp / 64 values per lane; the columns past d are zero and stay zero)
    int64_t* doc_ptr = nullptr; int32_t* words = nullptr;
    uint32_t *sample_int = nullptr, *cum_table = nullptr;
    float *dv = nullptr, *wv = nullptr, *syn1neg = nullptr;
    int64_t* order = nullptr; double* progress = nullptr;
    double* d_loss = nullptr;     // [sum of -log terms, number of terms]
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    uint64_t seed = 0;
    std::string err;
};
static thread_local std::string g_d2v_create_error;

#define DCHK(h, call) do { hipError_t _s = (call); if (_s != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(_s); return NTF_EHIP; } } while (0)
#define DFAIL(h, code, msg) do { (h)->err = (msg); return (code); } while (0)

extern "C" const char* ntf_d2v_last_error(const ntf_d2v* h) { return h ? h->err.c_str() : g_d2v_create_error.c_str(); }

namespace {
constexpr int D2V_RING = 1024;      // kept words of a document a wave holds at a time: a ring filled on demand, 64 raw words per step, just ahead of the window it is read through
constexpr int D2V_MAX_DOC = 10000;  // gensim's cap (doc2vec_inner.pyx MAX_DOCUMENT_LEN): the words of a document that survive the subsampling are collected up to this many
constexpr float D2V_MAX_EXP = 6.f;
enum { SLOT_KEEP = 0, SLOT_WINDOW = 1, SLOT_NEG0 = 2, SLOT_NEG1 = 3 };

struct D2vArgs {
    int64_t n_docs, n_vocab; int d, window, negative, serial;
    const int64_t* doc_ptr; const int32_t* words; const uint32_t *sample_int, *cum_table; const int64_t* order; const double* progress;
    float *dv, *wv, *syn1neg;
    double alpha_start, alpha_end;
    uint32_t k0, k1;
    double* loss;
};

__device__ __forceinline__ uint4 d2v_draw(const D2vArgs& a, int64_t doc, int pos, int unit, int slot) {
    return philox4x32(make_uint4((uint32_t)doc, (uint32_t)((uint64_t)doc >> 32), ((uint32_t)pos << 8) | (uint32_t)unit, (uint32_t)slot), make_uint2(a.k0, a.k1));
}
// EXP_TABLE lookup of doc2vec_inner.pyx (the table is built in float32: entry i = sigmoid((i / 1000 * 2 - 1) * 6))
__device__ __forceinline__ float d2v_sigmoid_table(float f) {
    const int i = (int)((f + D2V_MAX_EXP) * (1000.f / D2V_MAX_EXP / 2.f));
    const float e = expf(((float)i / 1000.f * 2.f - 1.f) * D2V_MAX_EXP);
    return e / (e + 1.f);
}
// rows other waves add to: read past the CU's vector cache
__device__ __forceinline__ float d2v_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int D2V_TOP = 1024;       // cumulative table, first level: the last entry of each of <= 1024 equal buckets, in LDS (round 5: the bisection in memory was 13-16 dependent round trips a position;
                                    // now ~5: PV-DM 61.5 -> 55.2 ms a pass.  Fetching a unit's six syn1neg rows up front instead of one by one was tried with it: 95 ms - the registers cost two waves per SIMD)

// the negative-sampling unit: input x (registers), predicted word `word`; returns `work` in w[], adds g x to the rows of syn1neg it touches
template <int NV>
__device__ __forceinline__ void d2v_unit(const D2vArgs& a, const uint32_t* __restrict__ top, int top_n, int top_s, const float (&x)[NV], int word, float alpha, int64_t doc, int pos, int unit,
                                         int lane, float (&w)[NV], float& lsum, float& lcnt) {
#pragma unroll
    for (int k = 0; k < NV; ++k) w[k] = 0.f;
    // lane k < 8 holds draw k: two Philox words of four draws each
    const uint4 r4 = d2v_draw(a, doc, pos, unit, SLOT_NEG0 + ((lane >> 2) & 1));
    const uint32_t rk = (lane & 3) == 0 ? r4.x : (lane & 3) == 1 ? r4.y : (lane & 3) == 2 ? r4.z : r4.w;
    // bisect_left(cum_table, r % cum_table[-1]): the bucket through the LDS level (the first bucket whose last entry is >= v holds the answer), then inside it
    int tgt;
    {
        const uint32_t v = rk % a.cum_table[a.n_vocab - 1];
        int blo = 0, bhi = top_n;
        while (blo < bhi) { const int mid = (blo + bhi) >> 1; if (top[mid] < v) blo = mid + 1; else bhi = mid; }
        int64_t lo = (int64_t)blo * top_s, hi = min((int64_t)(blo + 1) * top_s, a.n_vocab);
        if (blo >= top_n) { lo = a.n_vocab; hi = a.n_vocab; }        // (v is below cum_table[-1]: never taken)
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.cum_table[mid] < v) lo = mid + 1; else hi = mid; }
        tgt = (int)lo;
    }
    for (int k = 0; k <= a.negative; ++k) {
        const int t = __builtin_amdgcn_readfirstlane(k == 0 ? word : __shfl(tgt, k - 1, 64));      // (wave-uniform: row addresses in scalar registers)
        if (k > 0 && t == word) continue;
        const float label = k == 0 ? 1.f : 0.f;
        float* row = a.syn1neg + (int64_t)t * a.d;
        float rv[NV], f = 0.f;
#pragma unroll
        for (int q = 0; q < NV; ++q) { rv[q] = d2v_ld(row + lane + 64 * q); f += x[q] * rv[q]; }
        f = wave_reduce_sum(f);
        if (f <= -D2V_MAX_EXP || f >= D2V_MAX_EXP) continue;
        const float s = d2v_sigmoid_table(f);
        lsum -= logf(fmaxf(label != 0.f ? s : 1.f - s, 1e-30f)); lcnt += 1.f;
        const float g = (label - s) * alpha;
#pragma unroll
        for (int q = 0; q < NV; ++q) { w[q] += g * rv[q]; unsafeAtomicAdd(row + lane + 64 * q, g * x[q]); }
    }
}

// DW > 0 (round 5): PV-DM with window == DW adds to a word vector ONCE per kept position instead of once per (position, window word).  A word at kept position m is a
// window word of positions m - DW .. m + DW: a wave that walks the document front to back holds the sums of its own pending additions to the rows of positions
// i - DW .. i + DW in registers (pend[2 DW + 1]: NV values a lane, shifted by one per position), reads a window row as memory + pending - exactly what the row would hold
// had the additions been made (gensim's sequential order inside a document; other waves' additions arrive through memory as before) - and makes ONE atomic row add when the
// position leaves the reach of every later window.  Per position: 1 + 6 atomic row adds instead of ~5 + 6 (the kernel runs at 0.8 of the chip's f32 atomic rate).  Two kept
// positions inside one reach that hold the SAME word would each miss the other's pending sum: the wave then flushes what it holds and finishes the document the plain way.
// (occupancy: the plain kernel fits six waves per SIMD in 78 registers when asked to - PV-DBOW 268 -> 253 ms a pass; the deferred one needs its 127: held to five or six it spills and loses 20 %)
template <int NV, int DW = 0>
__global__ __launch_bounds__(256, DW > 0 ? 1 : (NV <= 2 ? 6 : 5)) void k_d2v_epoch(D2vArgs a, int dm) {
    __shared__ int s_kept[4][D2V_RING];
    __shared__ uint32_t s_top[D2V_TOP];
    const int lane = threadIdx.x & 63, wv_id = threadIdx.x >> 6;
    int* kept = s_kept[wv_id];
    const int top_s = (int)((a.n_vocab + D2V_TOP - 1) / D2V_TOP), top_n = (int)((a.n_vocab + top_s - 1) / top_s);     // buckets of top_s entries; top[k] = the last entry of bucket k
    for (int k = threadIdx.x; k < top_n; k += blockDim.x) s_top[k] = a.cum_table[min((int64_t)(k + 1) * top_s, a.n_vocab) - 1];
    __syncthreads();
    const int64_t first = a.serial ? 0 : (int64_t)blockIdx.x * 4 + wv_id;
    const int64_t stride = a.serial ? 1 : (int64_t)gridDim.x * 4;
    if (a.serial && (blockIdx.x != 0 || wv_id != 0)) return;
    float lsum = 0.f, lcnt = 0.f;
    for (int64_t rank = first; rank < a.n_docs; rank += stride) {
        const int64_t doc = a.order ? a.order[rank] : rank;
        const float alpha = (float)(a.alpha_start - (a.alpha_start - a.alpha_end) * (a.progress ? a.progress[rank] : (double)rank / (double)a.n_docs));
        const int64_t p0 = a.doc_ptr[doc];
        const int L = (int)(a.doc_ptr[doc + 1] - p0);
        // ---- words that survive the frequent-word subsampling (sample_int >= draw), in order: compacted by ballot into a RING, 64 raw words per step, filled just
        //      far enough ahead of position i for its window (i + window) - a document of any length needs 1 024 slots (round 3 held the whole kept list in LDS and
        //      silently cut documents at 1 024 kept words; gensim's cap is 10 000: embtype member / skillmember on gith, uspt)
        int K = 0, base = 0;               // kept words compacted so far (<= D2V_MAX_DOC), raw words scanned so far
        float dreg[NV];
        float* drow = a.dv + doc * a.d;
#pragma unroll
        for (int q = 0; q < NV; ++q) dreg[q] = drow[lane + 64 * q];
        constexpr int NPEND = DW > 0 ? 2 * DW + 1 : 1;
        float pend[NPEND][NV];             // DW > 0: index j <-> kept position i - DW + j
        bool defer = DW > 0;
#pragma unroll
        for (int j = 0; j < NPEND; ++j)
#pragma unroll
            for (int q = 0; q < NV; ++q) pend[j][q] = 0.f;
        auto flush_row = [&](int m, const float (&v)[NV]) {
            float* r = a.wv + (int64_t)__builtin_amdgcn_readfirstlane(kept[m & (D2V_RING - 1)]) * a.d;
#pragma unroll
            for (int q = 0; q < NV; ++q) unsafeAtomicAdd(r + lane + 64 * q, v[q]);
        };
        int i = 0;
        for (;; ++i) {
            while (K < i + a.window + 1 && base < L && K < D2V_MAX_DOC) {
                const int p = base + lane;
                int wd = 0; bool keep = false;
                if (p < L) { wd = a.words[p0 + p]; keep = a.sample_int[wd] >= d2v_draw(a, doc, p, 0, SLOT_KEEP).x; }
                const unsigned long long m = __ballot(keep);
                const int at = K + __popcll(m & ((1ull << lane) - 1ull));
                if (keep && at < D2V_MAX_DOC) kept[at & (D2V_RING - 1)] = wd;         // (slots older than i - window are dead: K - (i - window) <= 2 window + 64 < D2V_RING)
                K = min(K + (int)__popcll(m), D2V_MAX_DOC);
                base += 64;
                __builtin_amdgcn_wave_barrier();
            }
            if (i >= K) break;
            const int b = (int)(d2v_draw(a, doc, i, 0, SLOT_WINDOW).x % (uint32_t)a.window), word = __builtin_amdgcn_readfirstlane(kept[i & (D2V_RING - 1)]);
            // (K is final here whenever it bounds the window: K < i + window + 1 only once every raw word has been scanned or the cap is reached)
            const int lo = max(0, i - a.window + b), hi = min(K, i + a.window + 1 - b);
            float work[NV];
            if (DW > 0 && dm) {
                if (defer) {
                    // the same word at two kept positions of this reach?  lane j holds the word of position i - DW + j (a different negative number where there is none)
                    const int mj = i - DW + lane;
                    const int wj = (lane < NPEND && mj >= 0 && mj < K) ? kept[mj & (D2V_RING - 1)] : -1 - lane;
                    bool dup = false;
#pragma unroll
                    for (int o = 1; o < NPEND; ++o) dup |= (__shfl(wj, (lane + o) & 63, 64) == wj) && lane + o < NPEND;
                    if (__ballot(dup) != 0ull) {
#pragma unroll
                        for (int j = 0; j < NPEND; ++j) {
                            const int m = i - DW + j;
                            if (m >= 0 && m < K) flush_row(m, pend[j]);
#pragma unroll
                            for (int q = 0; q < NV; ++q) pend[j][q] = 0.f;
                        }
                        defer = false;
                    }
                }
                float l1[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) l1[q] = dreg[q];
#pragma unroll
                for (int j = 0; j < NPEND; ++j) {
                    const int m = i - DW + j;
                    if (j == DW || m < lo || m >= hi) continue;
                    const float* r = a.wv + (int64_t)__builtin_amdgcn_readfirstlane(kept[m & (D2V_RING - 1)]) * a.d;
#pragma unroll
                    for (int q = 0; q < NV; ++q) l1[q] += d2v_ld(r + lane + 64 * q) + pend[j][q];
                }
                const float inv = 1.f / (float)(hi - lo);               // (hi - lo - 1) window words + the doc tag
#pragma unroll
                for (int q = 0; q < NV; ++q) l1[q] *= inv;
                d2v_unit<NV>(a, s_top, top_n, top_s, l1, word, alpha, doc, i, 0, lane, work, lsum, lcnt);
#pragma unroll
                for (int q = 0; q < NV; ++q) dreg[q] += work[q];
#pragma unroll
                for (int j = 0; j < NPEND; ++j) {
                    const int m = i - DW + j;
                    if (j == DW || m < lo || m >= hi) continue;
                    if (defer) {
#pragma unroll
                        for (int q = 0; q < NV; ++q) pend[j][q] += work[q];
                    } else flush_row(m, work);
                }
                // position i - DW is in no later window: its row takes the sum; every slot moves one down
                if (defer && i - DW >= 0) flush_row(i - DW, pend[0]);
#pragma unroll
                for (int j = 0; j + 1 < NPEND; ++j)
#pragma unroll
                    for (int q = 0; q < NV; ++q) pend[j][q] = pend[j + 1][q];
#pragma unroll
                for (int q = 0; q < NV; ++q) pend[NPEND - 1][q] = 0.f;
            } else if (dm) {
                float l1[NV];
#pragma unroll
                for (int q = 0; q < NV; ++q) l1[q] = dreg[q];
                for (int m = lo; m < hi; ++m) {
                    if (m == i) continue;
                    const float* r = a.wv + (int64_t)__builtin_amdgcn_readfirstlane(kept[m & (D2V_RING - 1)]) * a.d;
#pragma unroll
                    for (int q = 0; q < NV; ++q) l1[q] += d2v_ld(r + lane + 64 * q);
                }
                const float inv = 1.f / (float)(hi - lo);               // (hi - lo - 1) window words + the doc tag
#pragma unroll
                for (int q = 0; q < NV; ++q) l1[q] *= inv;
                d2v_unit<NV>(a, s_top, top_n, top_s, l1, word, alpha, doc, i, 0, lane, work, lsum, lcnt);
#pragma unroll
                for (int q = 0; q < NV; ++q) dreg[q] += work[q];
                for (int m = lo; m < hi; ++m) {
                    if (m == i) continue;
                    float* r = a.wv + (int64_t)__builtin_amdgcn_readfirstlane(kept[m & (D2V_RING - 1)]) * a.d;
#pragma unroll
                    for (int q = 0; q < NV; ++q) unsafeAtomicAdd(r + lane + 64 * q, work[q]);
                }
            } else {
                int u = 0;
                for (int m = lo; m < hi; ++m) {
                    if (m == i) continue;
                    float* r = a.wv + (int64_t)__builtin_amdgcn_readfirstlane(kept[m & (D2V_RING - 1)]) * a.d;
                    float x[NV];
#pragma unroll
                    for (int q = 0; q < NV; ++q) x[q] = d2v_ld(r + lane + 64 * q);
                    d2v_unit<NV>(a, s_top, top_n, top_s, x, word, alpha, doc, i, 1 + u, lane, work, lsum, lcnt);
#pragma unroll
                    for (int q = 0; q < NV; ++q) unsafeAtomicAdd(r + lane + 64 * q, work[q]);
                    ++u;
                }
                d2v_unit<NV>(a, s_top, top_n, top_s, dreg, word, alpha, doc, i, 0, lane, work, lsum, lcnt);
#pragma unroll
                for (int q = 0; q < NV; ++q) dreg[q] += work[q];
            }
        }
        if (DW > 0 && dm && defer) {      // the document has ended at i = K: index j still stands for position i - DW + j; those below K hold sums
#pragma unroll
            for (int j = 0; j < DW; ++j) {
                const int m = i - DW + j;
                if (m >= 0 && m < K) flush_row(m, pend[j]);
            }
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) drow[lane + 64 * q] = dreg[q];
        __builtin_amdgcn_wave_barrier();
    }
    if (a.loss && lane == 0 && lcnt > 0.f) { atomicAdd(a.loss, (double)lsum); atomicAdd(a.loss + 1, (double)lcnt); }
}

template <typename T> int dalloc(ntf_d2v* h, T** p, int64_t n) {
    *p = nullptr;
    if (n <= 0) return NTF_OK;
    if (hipMalloc((void**)p, (size_t)n * sizeof(T)) != hipSuccess) { h->err = "hipMalloc failed"; return NTF_ENOMEM; }
    return NTF_OK;
}
void d2v_key(uint64_t seed, uint64_t epoch, uint32_t& k0, uint32_t& k1) {
    uint64_t x = seed ^ (epoch * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    k0 = (uint32_t)x; k1 = (uint32_t)(x >> 32);
}
}  // namespace

extern "C" void ntf_d2v_destroy(ntf_d2v* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->st) hipStreamSynchronize(h->st);
    for (void* p : {(void*)h->doc_ptr, (void*)h->words, (void*)h->sample_int, (void*)h->cum_table, (void*)h->dv, (void*)h->wv, (void*)h->syn1neg, (void*)h->order, (void*)h->progress, (void*)h->d_loss})
        if (p) hipFree(p);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

extern "C" int ntf_d2v_create(int device, int64_t n_docs, int64_t n_vocab, int32_t d, const int64_t* doc_ptr, const int32_t* words, const uint32_t* sample_int,
                              const uint32_t* cum_table, const float* init_wv, const float* init_dv, uint64_t seed, ntf_d2v** out) {
    if (!out) { g_d2v_create_error = "out is NULL"; return NTF_EINVAL; }
    *out = nullptr;
    if (n_docs < 1 || n_vocab < 1 || d < 1 || d > 256 || !doc_ptr || !words || !sample_int || !cum_table || !init_wv || !init_dv) {
        g_d2v_create_error = "d2v: need n_docs >= 1, n_vocab >= 1, 1 <= d <= 256, the documents, the vocabulary tables and the initial vectors"; return NTF_EINVAL; }
    if (doc_ptr[0] != 0) { g_d2v_create_error = "d2v: doc_ptr[0] != 0"; return NTF_EINVAL; }
    for (int64_t i = 0; i < n_docs; ++i) if (doc_ptr[i + 1] < doc_ptr[i]) { g_d2v_create_error = "d2v: doc_ptr not monotone"; return NTF_EINVAL; }
    const int64_t nw = doc_ptr[n_docs];
    for (int64_t p = 0; p < nw; ++p) if (words[p] < 0 || words[p] >= n_vocab) { g_d2v_create_error = "d2v: word index out of the vocabulary"; return NTF_EINVAL; }
    for (int64_t v = 1; v < n_vocab; ++v) if (cum_table[v] < cum_table[v - 1]) { g_d2v_create_error = "d2v: cum_table not monotone"; return NTF_EINVAL; }
    if (cum_table[n_vocab - 1] == 0) { g_d2v_create_error = "d2v: empty cum_table"; return NTF_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_d2v_create_error = "no such HIP device (there is no CPU fallback)"; return NTF_EHIP; }
    hipSetDevice(device);
    ntf_d2v* h = new ntf_d2v();
    h->device = device; h->n_docs = n_docs; h->n_vocab = n_vocab; h->n_words = nw; h->d = d; h->dp = (d + 63) / 64 * 64; h->seed = seed;
    const int64_t dp = h->dp;
    int rc = NTF_OK;
    auto A = [&](int r) { if (rc == NTF_OK) rc = r; };
    if (hipStreamCreate(&h->st) != hipSuccess) { g_d2v_create_error = "hipStreamCreate failed"; delete h; return NTF_EHIP; }
    A(dalloc(h, &h->doc_ptr, n_docs + 1)); A(dalloc(h, &h->words, std::max<int64_t>(nw, 1))); A(dalloc(h, &h->sample_int, n_vocab)); A(dalloc(h, &h->cum_table, n_vocab));
    A(dalloc(h, &h->dv, n_docs * dp)); A(dalloc(h, &h->wv, n_vocab * dp)); A(dalloc(h, &h->syn1neg, n_vocab * dp)); A(dalloc(h, &h->order, n_docs)); A(dalloc(h, &h->progress, n_docs)); A(dalloc(h, &h->d_loss, 2));
    if (rc != NTF_OK) { g_d2v_create_error = h->err; ntf_d2v_destroy(h); return rc; }
    hipMemcpy(h->doc_ptr, doc_ptr, (n_docs + 1) * 8, hipMemcpyHostToDevice);
    if (nw) hipMemcpy(h->words, words, nw * 4, hipMemcpyHostToDevice);
    hipMemcpy(h->sample_int, sample_int, n_vocab * 4, hipMemcpyHostToDevice);
    hipMemcpy(h->cum_table, cum_table, n_vocab * 4, hipMemcpyHostToDevice);
    // any vector size (the reference's CI trains d9 tables, data.embedding.d is free): rows of d floats at a stride of dp, the pad columns zero.  A zero column of
    // every table stays zero through every update (each update is a multiple of another table's same column), and adds nothing to a dot product
    if (dp != d) { hipMemset(h->dv, 0, (size_t)n_docs * dp * 4); hipMemset(h->wv, 0, (size_t)n_vocab * dp * 4); }
    hipMemcpy2D(h->dv, (size_t)dp * 4, init_dv, (size_t)d * 4, (size_t)d * 4, (size_t)n_docs, hipMemcpyHostToDevice);
    hipMemcpy2D(h->wv, (size_t)dp * 4, init_wv, (size_t)d * 4, (size_t)d * 4, (size_t)n_vocab, hipMemcpyHostToDevice);
    hipMemsetAsync(h->syn1neg, 0, (size_t)n_vocab * dp * 4, h->st);
    if (hipStreamSynchronize(h->st) != hipSuccess) { g_d2v_create_error = "device initialisation failed"; ntf_d2v_destroy(h); return NTF_EHIP; }
    *out = h;
    return NTF_OK;
}

extern "C" int ntf_d2v_train_epoch(ntf_d2v* h, int32_t dm, int32_t window, int32_t negative, double alpha_start, double alpha_end, uint64_t epoch, int32_t serial,
                                   const int64_t* order, const double* progress, double* mean_loss, double* device_ms) {
    if (!h || window < 1 || window > 255 || negative < 0 || negative > 8 || (dm != 0 && dm != 1)) return NTF_EINVAL;
    if (!dm && 2 * window > 254) return NTF_EINVAL;
    DCHK(h, hipSetDevice(h->device));
    if (order) {
        std::vector<char> seen((size_t)h->n_docs, 0);
        for (int64_t i = 0; i < h->n_docs; ++i) { if (order[i] < 0 || order[i] >= h->n_docs || seen[(size_t)order[i]]) DFAIL(h, NTF_EINVAL, "d2v: order is not a permutation of the documents"); seen[(size_t)order[i]] = 1; }
        DCHK(h, hipMemcpyAsync(h->order, order, h->n_docs * 8, hipMemcpyHostToDevice, h->st));
        DCHK(h, hipStreamSynchronize(h->st));
    }
    if (progress) {
        for (int64_t i = 0; i < h->n_docs; ++i) if (!(progress[i] >= 0.0 && progress[i] <= 1.0)) DFAIL(h, NTF_EINVAL, "d2v: progress outside [0, 1]");
        DCHK(h, hipMemcpyAsync(h->progress, progress, h->n_docs * 8, hipMemcpyHostToDevice, h->st));
        DCHK(h, hipStreamSynchronize(h->st));
    }
    DCHK(h, hipMemsetAsync(h->d_loss, 0, 16, h->st));
    D2vArgs a;
    a.n_docs = h->n_docs; a.n_vocab = h->n_vocab; a.d = h->dp; a.window = window; a.negative = negative; a.serial = serial ? 1 : 0;
    a.doc_ptr = h->doc_ptr; a.words = h->words; a.sample_int = h->sample_int; a.cum_table = h->cum_table; a.order = order ? h->order : nullptr; a.progress = progress ? h->progress : nullptr;
    a.dv = h->dv; a.wv = h->wv; a.syn1neg = h->syn1neg; a.alpha_start = alpha_start; a.alpha_end = alpha_end; a.loss = mean_loss ? h->d_loss : nullptr;
    d2v_key(h->seed, epoch, a.k0, a.k1);
    // parallel: enough waves to fill the chip several times over, each striding through the documents
    const int64_t want = (h->n_docs + 3) / 4;
    const dim3 grid(serial ? 1u : (unsigned)std::min<int64_t>(want, 256 * 16)), block(serial ? 64 : 256);
    if (device_ms) { if (!h->ev0) { DCHK(h, hipEventCreate(&h->ev0)); DCHK(h, hipEventCreate(&h->ev1)); } DCHK(h, hipEventRecord(h->ev0, h->st)); }
    // PV-DM at the reference's window (src/mdl/emb/__config__.yaml: w = 5): word-vector additions deferred to one per kept position (k_d2v_epoch<.., 5>); NTF_D2V_DEFER=0: the plain kernel
    static const bool defer_ok = !(getenv("NTF_D2V_DEFER") && atoi(getenv("NTF_D2V_DEFER")) == 0);
    if (dm && window == 5 && defer_ok && h->dp >= 128) {     // (d = 64: the rows are 256 B and the plain kernel's shorter chain wins - 45.7 against 49.4 ms; d = 128 / 192 / 256: 55 / 78 / 98 against 66 / 102 / 133)
        switch (h->dp / 64) {
            case 2: hipLaunchKernelGGL((k_d2v_epoch<2, 5>), grid, block, 0, h->st, a, dm); break;
            case 3: hipLaunchKernelGGL((k_d2v_epoch<3, 5>), grid, block, 0, h->st, a, dm); break;
            default: hipLaunchKernelGGL((k_d2v_epoch<4, 5>), grid, block, 0, h->st, a, dm); break;
        }
    } else
    switch (h->dp / 64) {
        case 1: hipLaunchKernelGGL(k_d2v_epoch<1>, grid, block, 0, h->st, a, dm); break;
        case 2: hipLaunchKernelGGL(k_d2v_epoch<2>, grid, block, 0, h->st, a, dm); break;
        case 3: hipLaunchKernelGGL(k_d2v_epoch<3>, grid, block, 0, h->st, a, dm); break;
        default: hipLaunchKernelGGL(k_d2v_epoch<4>, grid, block, 0, h->st, a, dm); break;
    }
    if (device_ms) DCHK(h, hipEventRecord(h->ev1, h->st));
    hipError_t s = hipGetLastError();
    if (s != hipSuccess) DFAIL(h, NTF_EHIP, std::string("d2v kernel launch: ") + hipGetErrorString(s));
    DCHK(h, hipStreamSynchronize(h->st));
    if (device_ms) { float ms = 0.f; DCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1)); *device_ms = ms; }
    if (mean_loss) {
        double l[2] = {0, 0};
        DCHK(h, hipMemcpy(l, h->d_loss, 16, hipMemcpyDeviceToHost));
        *mean_loss = l[1] > 0 ? l[0] / l[1] : 0.0;
    }
    return NTF_OK;
}

extern "C" int ntf_d2v_get(ntf_d2v* h, int what, float* host) {   // what: 0 = doc vectors [n_docs, d], 1 = word vectors [n_vocab, d], 2 = syn1neg [n_vocab, d]
    if (!h || !host || what < 0 || what > 2) return NTF_EINVAL;
    DCHK(h, hipSetDevice(h->device));
    DCHK(h, hipStreamSynchronize(h->st));
    const float* src = what == 0 ? h->dv : what == 1 ? h->wv : h->syn1neg;
    const int64_t rows = what == 0 ? h->n_docs : h->n_vocab;
    DCHK(h, hipMemcpy2D(host, (size_t)h->d * 4, src, (size_t)h->dp * 4, (size_t)h->d * 4, (size_t)rows, hipMemcpyDeviceToHost));
    return NTF_OK;
}

extern "C" int ntf_d2v_set(ntf_d2v* h, int what, const float* host) {   // resume from a saved table (same `what` as ntf_d2v_get)
    if (!h || !host || what < 0 || what > 2) return NTF_EINVAL;
    DCHK(h, hipSetDevice(h->device));
    DCHK(h, hipStreamSynchronize(h->st));
    float* dst = what == 0 ? h->dv : what == 1 ? h->wv : h->syn1neg;
    const int64_t rows = what == 0 ? h->n_docs : h->n_vocab;
    DCHK(h, hipMemcpy2D(dst, (size_t)h->dp * 4, host, (size_t)h->d * 4, (size_t)h->d * 4, (size_t)rows, hipMemcpyHostToDevice));
    return NTF_OK;
}
