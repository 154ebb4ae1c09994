// HBM-bound kernels of the CVAE train step for gfx950: BatchNorm (+LeakyReLU,
// +residual) forward/backward, column reductions, reparameterisation + KL,
// dropout, bilinear-upsample + reflection pad and their transposes, tanh + MSE,
// AdamW, layout helpers.  All of them move 16 bytes per lane per access on NHWC
// rows and reduce with per-thread accumulators -> LDS -> per-block partials ->
// a small finalize kernel (fixed summation order: results are run-to-run stable).
#include "common.h"

namespace {

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : LRELU_SLOPE * v; }

// ---- column reductions over rows[M][ld] ----------------------------------------------
// Thread layout: chunk = tid % CC handles 8 channels, rl = tid / CC strides the rows.
enum { RED_STATS = 0, RED_BNBWD = 1, RED_SUM = 2, RED_BNBWD_X = 3, RED_BNBWD_PLAIN = 4 };
// _X: gate recomputed from x, y not read; _PLAIN: dy already carries the activation's derivative

struct RedArgs {
    const void* x;       // STATS/SUM: input rows; BNBWD: conv output x
    const void* dy;      // BNBWD
    const void* y;       // BNBWD: activation output (sign source), or null: the sign of gamma*xhat+beta
    const float* stats;  // BNBWD: mean[C], invstd[C]
    const float* gamma;  // BNBWD without y
    const float* beta;
    float* ws;           // partials [2][nblk][C]
    float* shift_out;    // STATS: [C] receives the shift the sums are taken about (row 0), or null
    int M, C, ld, nblk;
    int Cb;              // columns handled per blockIdx.y (== C unless C > 1024; the last block may be ragged)
    // Totals in the same launch (fin != FIN_NONE): the block partials meet in two levels behind arrival counters
    // (groups of FIN_GROUP consecutive blocks, then the groups), each level summed in index order by whichever
    // block arrives last, so the totals do not depend on the arrival order and no second kernel is needed.
    int fin;             // FIN_NONE, FIN_STATS (mean / invstd / running statistics), FIN_SUMS (out0 / out1)
    double* ws2;         // level-2 partials [2][ngroups][C]
    unsigned* tickets;   // [gridDim.y][ngroups + 1] arrival counters, zero before and after every launch
    float* out0; float* out1; int md, ma, mb;                      // FIN_SUMS: out0[(c / md) * ma + (c % md) * mb], out1[c]
    float* stats_out; float* running_mean; float* running_var; long long* nbt;   // FIN_STATS
};
enum { FIN_NONE = 0, FIN_STATS = 1, FIN_SUMS = 2 };
constexpr int FIN_GROUP = 16;

// Loads that read past the per-XCD L2 (sc1: coherent at agent scope).  The compiler completes every atomic load
// before it issues the next one; these are issued back to back instead and completed together by ld_agent_wait,
// which also carries the values, so that nothing uses one before the wait.
__device__ __forceinline__ float ld_agent(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ double ld_agent(const double* p) {
    double v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <typename V, int N>
__device__ __forceinline__ void ld_agent_wait(V (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[i]) : : "memory");
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) colreduce(const RedArgs a) {
    const int CC = a.Cb / 8;
    const int RL = blockDim.x / CC;
    const int chunk = threadIdx.x % CC, rl = threadIdx.x / CC;
    const int c_off = blockIdx.y * a.Cb;
    const bool col_ok = c_off + chunk * 8 < a.C;          // C is a multiple of 8: a chunk is whole or absent
    __shared__ float sm[2][256][8];
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
    float mean[8], istd[8], sc[8], sh[8];
    constexpr bool from_x = MODE == RED_BNBWD_X;
    constexpr bool plain = MODE == RED_BNBWD_PLAIN;
    if (MODE == RED_BNBWD || from_x || plain) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = col_ok ? c_off + chunk * 8 + j : 0;
            mean[j] = a.stats[c]; istd[j] = a.stats[a.C + c];
            sc[j] = sh[j] = 0.f;
            if (from_x) { sc[j] = a.gamma[c] * istd[j]; sh[j] = a.beta[c] - mean[j] * sc[j]; }   // as bn_act_fwd
        }
    }
    // STATS sums are taken about row 0 (sum of (x - x0), sum of (x - x0)^2): E[x^2] - mean^2 over raw values loses
    // log2(mean^2 / var) bits, which a small batch behind a large mean turns into percent-level errors of the
    // BatchNorm backward pass ((1 - xhat^2) is eps / var there)
    float shift[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) shift[j] = 0.f;
    if (MODE == RED_STATS && a.shift_out && col_ok) {
        load8<T>((const T*)a.x + c_off + chunk * 8, shift);
        if (blockIdx.x == 0 && rl == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __hip_atomic_store(&a.shift_out[c_off + chunk * 8 + j], shift[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (rl < RL && col_ok) {
        // two rows per trip: their loads are issued together (twice the bytes in flight per thread),
        // the sums are taken in row order, so the result does not depend on the pairing
        const int NQ = (MODE == RED_BNBWD) ? 3 : ((MODE == RED_BNBWD_X || plain) ? 2 : 1);
        const int64_t stride = (int64_t)gridDim.x * RL;
        auto fetch = [&](int64_t m, float (&v)[8], float (&d)[8], float (&o)[8]) {
            const size_t off = (size_t)m * a.ld + c_off + chunk * 8;
            load8<T>((const T*)a.x + off, v);
            if (NQ >= 2) load8<T>((const T*)a.dy + off, d);
            if (NQ >= 3) load8<T>((const T*)a.y + off, o);
        };
        auto add = [&](float (&v)[8], float (&d)[8], float (&o)[8]) {
            if (MODE == RED_STATS) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d0 = v[j] - shift[j]; s0[j] += d0; s1[j] += d0 * d0; }
            } else if (MODE == RED_SUM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) s0[j] += v[j];
            } else {
                if (from_x) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = v[j] * sc[j] + sh[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g = (plain || o[j] > 0.f) ? d[j] : LRELU_SLOPE * d[j];
                    s0[j] += g;
                    s1[j] += g * (v[j] - mean[j]) * istd[j];
                }
            }
        };
        int64_t m = (int64_t)blockIdx.x * RL + rl;
        float va[8], da[8], oa[8], vb[8], db[8], ob[8];
        for (; m + stride < a.M; m += 2 * stride) {
            fetch(m, va, da, oa);
            fetch(m + stride, vb, db, ob);
            add(va, da, oa);
            add(vb, db, ob);
        }
        if (m < a.M) {
            fetch(m, va, da, oa);
            add(va, da, oa);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { sm[0][threadIdx.x][j] = s0[j]; sm[1][threadIdx.x][j] = s1[j]; }
    __syncthreads();
    // thread t < 2*C sums column (t % C) of partial (t / C) over the row lanes
    for (int t = threadIdx.x; t < 2 * a.Cb; t += blockDim.x) {
        const int which = t / a.Cb, col = t % a.Cb;
        const int ch = col / 8, j = col % 8;
        float s = 0.f;
        for (int r = 0; r < RL; ++r) s += sm[which][r * CC + ch][j];
        if (c_off + col < a.C)
            __hip_atomic_store(&a.ws[((size_t)which * a.nblk + blockIdx.x) * a.C + c_off + col], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.fin == FIN_NONE) return;

    // ---- totals: the last block of each group of FIN_GROUP sums the group's rows, the last group the groups.
    // Everything that crosses workgroups here (partial rows, level-2 rows, the shift row, counters) is written and
    // read with agent-scope atomic accesses: those write through / read past the per-XCD L2, so no cache-wide
    // release / acquire fence is needed (one such fence costs ~10 us on this part, and four of them would sit on
    // the critical path of the last block).  A writer waits for its stores to be acknowledged before it counts
    // itself in; a reader loads only after it has seen the full count.
    __shared__ int role;
    const int ngroups = (a.nblk + FIN_GROUP - 1) / FIN_GROUP;
    const int grp = blockIdx.x / FIN_GROUP, g_beg = grp * FIN_GROUP;
    const int g_cnt = min(FIN_GROUP, a.nblk - g_beg);
    unsigned* tk = a.tickets + (size_t)blockIdx.y * (ngroups + 1);
    __builtin_amdgcn_s_waitcnt(0);                   // this thread's partial stores have been acknowledged ...
    __syncthreads();
    if (threadIdx.x == 0)                            // ... before the block is counted
        role = __hip_atomic_fetch_add(&tk[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(g_cnt - 1) ? 1 : 0;
    __syncthreads();
    if (!role) return;
    for (int t = threadIdx.x; t < 2 * a.Cb; t += blockDim.x) {
        const int which = t / a.Cb, col = t % a.Cb;
        if (c_off + col >= a.C) continue;
        const float* p = a.ws + ((size_t)which * a.nblk + g_beg) * a.C + c_off + col;
        float v[FIN_GROUP];
#pragma unroll
        for (int r = 0; r < FIN_GROUP; ++r) {        // all loads in flight together
            v[r] = 0.f;
            if (r < g_cnt) v[r] = ld_agent(p + (size_t)r * a.C);
        }
        ld_agent_wait(v);
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < FIN_GROUP; ++r) acc += (double)v[r];
        __hip_atomic_store(&a.ws2[((size_t)which * ngroups + grp) * a.C + c_off + col], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&tk[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);           // ready for the next launch
        role = __hip_atomic_fetch_add(&tk[ngroups], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ngroups - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!role) return;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&tk[ngroups], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.fin == FIN_STATS && a.nbt && blockIdx.y == 0) *a.nbt += 1;
    }
    for (int col = threadIdx.x; col < a.Cb; col += blockDim.x) {
        const int c = c_off + col;
        if (c >= a.C) continue;
        double s = 0.0, ss = 0.0;
        for (int g0 = 0; g0 < ngroups; g0 += 8) {
            double va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = vb[u] = 0.0;
                if (g0 + u < ngroups) {
                    va[u] = ld_agent(&a.ws2[(size_t)(g0 + u) * a.C + c]);
                    vb[u] = ld_agent(&a.ws2[((size_t)ngroups + g0 + u) * a.C + c]);
                }
            }
            ld_agent_wait(va);
            ld_agent_wait(vb);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += va[u]; ss += vb[u]; }
        }
        if (a.fin == FIN_STATS) {
            const double dm = s / a.M;
            const double mean = dm + (a.shift_out ? (double)__hip_atomic_load(&a.shift_out[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0);
            double var = ss / a.M - dm * dm;
            if (var < 0.0) var = 0.0;
            a.stats_out[c] = (float)mean;
            a.stats_out[a.C + c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
            if (a.running_mean) {
                const double unb = a.M > 1 ? var * ((double)a.M / (a.M - 1)) : var;
                a.running_mean[c] = (1.f - BN_MOMENTUM) * a.running_mean[c] + BN_MOMENTUM * (float)mean;
                a.running_var[c] = (1.f - BN_MOMENTUM) * a.running_var[c] + BN_MOMENTUM * (float)unb;
            }
        } else {
            a.out0[(c / a.md) * a.ma + (c % a.md) * a.mb] = (float)s;
            if (a.out1) a.out1[c] = (float)ss;
        }
    }
}

// Partial sums [2][nblk][C] -> per-channel totals.  One wave per channel (4 per block): every lane
// issues all of its loads (nblk <= 512: at most 8) before the first add, then the 64 lanes meet in
// a butterfly whose order is fixed, so the totals are run-to-run stable.
constexpr int FIN_CH = 4;                   // channels per 256-thread block
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void sum_partials(const float* __restrict__ ws, int nblk, int C, int c, double& s,
                                             double& ss, bool want_ss) {
    const int lane = threadIdx.x & 63;
    double a = 0.0, b = 0.0;
    if (c < C) {
        for (int k0 = 0; k0 < nblk; k0 += 512) {
            float va[8], vb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = k0 + lane + 64 * i;
                va[i] = k < nblk ? ws[(size_t)k * C + c] : 0.f;
                vb[i] = (want_ss && k < nblk) ? ws[((size_t)nblk + k) * C + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { a += va[i]; b += vb[i]; }
        }
    }
    s = wave_sum_d(a);
    ss = wave_sum_d(b);
}

// shift: [C] the value the sums were taken about, or null (raw sums).  It may alias stats + C (the
// statistics kernel parks it there): each channel's thread reads it before it writes the inverse deviation.
__global__ void __launch_bounds__(256) stats_finalize(const float* __restrict__ ws, int nblk, int M, int C,
                                                      float* stats, float* running_mean,
                                                      float* running_var, long long* nbt, const float* shift) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    double s, ss;
    sum_partials(ws, nblk, C, c, s, ss, true);
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    if ((threadIdx.x & 63) != 0 || c >= C) return;
    const double dm = s / M;
    const double mean = dm + (shift ? (double)shift[c] : 0.0);
    double var = ss / M - dm * dm;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    if (running_mean) {
        const double unb = M > 1 ? var * ((double)M / (M - 1)) : var;
        running_mean[c] = (1.f - BN_MOMENTUM) * running_mean[c] + BN_MOMENTUM * (float)mean;
        running_var[c] = (1.f - BN_MOMENTUM) * running_var[c] + BN_MOMENTUM * (float)unb;
    }
}

__global__ void eval_stats(const float* rm, const float* rv, int C, float* stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stats[c] = rm[c];
    stats[C + c] = 1.f / sqrtf(rv[c] + BN_EPS);
}

// sums partials; out0[map(c)] = sum of partial 0, out1[c] = sum of partial 1 (optional)
__global__ void __launch_bounds__(256) sum_finalize(const float* __restrict__ ws, int nblk, int C, float* out0,
                                                    float* out1, int md, int ma, int mb) {
    const int c = blockIdx.x * FIN_CH + (threadIdx.x >> 6);
    double s, t;
    sum_partials(ws, nblk, C, c, s, t, out1 != nullptr);
    if ((threadIdx.x & 63) != 0 || c >= C) return;
    out0[(c / md) * ma + (c % md) * mb] = (float)s;
    if (out1) out1[c] = (float)t;
}

// ---- BatchNorm + LeakyReLU apply -------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) bn_act_fwd(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                  int M, int C, int Cb, const float* __restrict__ stats,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta) {
    // Cb columns per blockIdx.y (== C unless C > 1024); `chunk` is the 8-column chunk within the row
    const int CC = Cb / 8, RL = blockDim.x / CC;
    const int chunk = blockIdx.y * CC + threadIdx.x % CC, rl = threadIdx.x / CC;
    if (rl >= RL || chunk * 8 >= C) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk * 8 + j;
        sc[j] = gamma[c] * stats[C + c];
        sh[j] = beta[c] - stats[c] * sc[j];
    }
    for (int64_t m = (int64_t)blockIdx.x * RL + rl; m < M; m += (int64_t)gridDim.x * RL) {
        const size_t off = (size_t)m * C + chunk * 8;
        float v[8];
        load8<T>(x + off, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
        if (res) {
            float r[8];
            load8<T>(res + off, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += r[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lrelu(v[j]);
        store8<T>(y + off, v);
    }
}

// GATE 0: sign of y; 1: sign of gamma*xhat + beta, recomputed as bn_act_fwd does; 2: none (dy is already gated)
template <typename T, int GATE>
__global__ void __launch_bounds__(256) bn_act_bwd_dx(const T* __restrict__ dy, const T* __restrict__ y,
                                                     const T* __restrict__ x, T* __restrict__ dx, int M, int C, int Cb,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta) {
    constexpr bool from_x = GATE == 1;
    const int CC = Cb / 8, RL = blockDim.x / CC;
    const int chunk = blockIdx.y * CC + threadIdx.x % CC, rl = threadIdx.x / CC;
    if (rl >= RL || chunk * 8 >= C) return;
    float mean[8], istd[8], k0[8], k1[8], k2[8], sh[8];
    const float invM = 1.f / (float)M;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = chunk * 8 + j;
        mean[j] = stats[c];
        istd[j] = stats[C + c];
        k0[j] = gamma[c] * istd[j];
        k1[j] = dbeta[c] * invM;
        k2[j] = dgamma[c] * invM;
        sh[j] = from_x ? beta[c] - mean[j] * k0[j] : 0.f;
    }
    for (int64_t m = (int64_t)blockIdx.x * RL + rl; m < M; m += (int64_t)gridDim.x * RL) {
        const size_t off = (size_t)m * C + chunk * 8;
        float d[8], o[8], v[8];
        load8<T>(dy + off, d);
        load8<T>(x + off, v);
        if (from_x) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = v[j] * k0[j] + sh[j];
        } else if (GATE == 0) {
            load8<T>(y + off, o);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = (GATE == 2 || o[j] > 0.f) ? d[j] : LRELU_SLOPE * d[j];
            const float xh = (v[j] - mean[j]) * istd[j];
            v[j] = k0[j] * (g - k1[j] - xh * k2[j]);
        }
        store8<T>(dx + off, v);
    }
}

// ---- heads and losses of the adversarial extension (discriminator / classifier on the encoder trunk)
// Rows are samples, F trunk features per row (row pitch ld >= F, padding columns written as 0).
// Everything is sized for a few thousand rows and F, y <= 1024: one wave or block per row, fixed
// summation orders, no atomics.

// Discriminator head: logit = <h, w> + b, BCE-with-logits against 1 for rows < n_real, 0 after.
template <typename T>
__global__ void __launch_bounds__(256) d_head_rows(const T* __restrict__ h, int ld, const float* __restrict__ w,
                                                   const float* __restrict__ b, int M, int F, int n_real, float gscale,
                                                   float* __restrict__ dlogit, float* __restrict__ loss_row,
                                                   T* __restrict__ dh) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    float acc = 0.f;
    for (int j = lane; j < F; j += 64) acc += to_f32(h[(size_t)row * ld + j]) * w[j];
    const float l = wave_sum(acc) + b[0];
    const float t = row < n_real ? 1.f : 0.f;
    const float sg = 1.f / (1.f + expf(-l));
    const float dl = (sg - t) * gscale / (float)M;
    if (lane == 0) {
        loss_row[row] = fmaxf(l, 0.f) - l * t + log1pf(expf(-fabsf(l)));
        dlogit[row] = dl;
    }
    for (int j = lane; j < ld; j += 64) dh[(size_t)row * ld + j] = from_f32<T>(j < F ? dl * w[j] : 0.f);
}
template <typename T>
__global__ void __launch_bounds__(1024) d_head_reduce(const T* __restrict__ h, int ld, const float* __restrict__ dlogit,
                                                      const float* __restrict__ loss_row, int M, int F,
                                                      float* __restrict__ dw, float* __restrict__ db,
                                                      float* __restrict__ loss_out) {
    const int j = threadIdx.x;
    if (j < F) {
        float s = 0.f;
        for (int r = 0; r < M; ++r) s += dlogit[r] * to_f32(h[(size_t)r * ld + j]);
        dw[j] = s;
    } else if (j == F) {
        float s = 0.f;
        for (int r = 0; r < M; ++r) s += dlogit[r];
        db[0] = s;
    } else if (j == F + 1) {
        float s = 0.f;
        for (int r = 0; r < M; ++r) s += loss_row[r];
        loss_out[0] = s / (float)M;
    }
}

// Classifier head: logits = h W^T + b ([y][F] weights), softmax cross-entropy over the first n_real
// rows (mean); the other rows contribute nothing and get a zero gradient.
template <typename T>
__global__ void __launch_bounds__(256) c_head_rows(const T* __restrict__ h, int ld, const float* __restrict__ W,
                                                   const float* __restrict__ b, const int* __restrict__ labels, int M,
                                                   int F, int y, int n_real, float gscale, float* __restrict__ dlogits,
                                                   float* __restrict__ loss_row, T* __restrict__ dh) {
    __shared__ float hrow[1024], lg[1024], red[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bool real = row < n_real;
    for (int j = tid; j < F; j += 256) hrow[j] = to_f32(h[(size_t)row * ld + j]);
    __syncthreads();
    float mx = -3.0e38f;
    for (int c = tid; c < y; c += 256) {
        float a = b[c];
        for (int j = 0; j < F; ++j) a += hrow[j] * W[(size_t)c * F + j];
        lg[c] = a;
        mx = fmaxf(mx, a);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float se = 0.f;
    for (int c = tid; c < y; c += 256) se += expf(lg[c] - mx);
    red[tid] = se;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    se = red[0];
    const int lab = real ? labels[row] : 0;
    const float lab_logit = lg[lab];
    const float k = gscale / (float)n_real;
    __syncthreads();                                  // everyone has read lg before it is overwritten
    for (int c = tid; c < y; c += 256) {
        const float pc = expf(lg[c] - mx) / se;
        const float d = real ? (pc - (c == lab ? 1.f : 0.f)) * k : 0.f;
        dlogits[(size_t)row * y + c] = d;
        lg[c] = d;
    }
    if (tid == 0) loss_row[row] = real ? (logf(se) + mx - lab_logit) : 0.f;
    __syncthreads();
    for (int j = tid; j < ld; j += 256) {
        float a = 0.f;
        if (real && j < F)
            for (int c = 0; c < y; ++c) a += lg[c] * W[(size_t)c * F + j];
        dh[(size_t)row * ld + j] = from_f32<T>(a);
    }
}
template <typename T>
__global__ void __launch_bounds__(1024) c_head_reduce(const T* __restrict__ h, int ld, const float* __restrict__ dlogits,
                                                      const float* __restrict__ loss_row, int F, int y, int n_real,
                                                      float* __restrict__ dW, float* __restrict__ db,
                                                      float* __restrict__ loss_out) {
    const int c = blockIdx.x, j = threadIdx.x;
    if (j < F) {
        float s = 0.f;
        for (int r = 0; r < n_real; ++r) s += dlogits[(size_t)r * y + c] * to_f32(h[(size_t)r * ld + j]);
        dW[(size_t)c * F + j] = s;
    } else if (j == F) {
        float s = 0.f;
        for (int r = 0; r < n_real; ++r) s += dlogits[(size_t)r * y + c];
        db[c] = s;
    } else if (j == F + 1 && c == 0) {
        float s = 0.f;
        for (int r = 0; r < n_real; ++r) s += loss_row[r];
        loss_out[0] = s / (float)n_real;
    }
}

// Mean feature matching: loss = 1/2 |mean(h[r0:r1]) - mean(h[f0:f1])|^2; the gradient goes to the
// rows [f0, f1) only (the first mean is a constant target), every other row of dh is zeroed.
template <typename T>
__global__ void __launch_bounds__(1024) feature_match(const T* __restrict__ h, int ld, int M, int F, int r0, int r1,
                                                      int f0, int f1, float gscale, float* __restrict__ loss_out,
                                                      T* __restrict__ dh) {
    __shared__ float red[1024];
    const int j = threadIdx.x;
    float d = 0.f;
    if (j < F) {
        float a = 0.f, c = 0.f;
        for (int r = r0; r < r1; ++r) a += to_f32(h[(size_t)r * ld + j]);
        for (int r = f0; r < f1; ++r) c += to_f32(h[(size_t)r * ld + j]);
        d = c / (float)(f1 - f0) - a / (float)(r1 - r0);
    }
    red[j] = d * d;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (j < o) red[j] += red[j + o]; __syncthreads(); }
    if (j == 0) loss_out[0] = 0.5f * red[0];
    if (j < ld) {
        const float g = d * gscale / (float)(f1 - f0);
        for (int r = 0; r < M; ++r) dh[(size_t)r * ld + j] = from_f32<T>((r >= f0 && r < f1) ? g : 0.f);
    }
}

// Per-class feature centres (the paper's L_GC): block c handles class c.  The centre of the REAL rows of
// class c is a moving average kept across steps (a constant target); the rows [f0, f1) (generated images)
// of class c are pulled towards it as a group: loss_c = 1/2 |mean_fake_c - centre_c|^2, and the gradient
// (mean_fake_c - centre_c) * gscale / n_fake_c is ADDED to those rows of dh.  Classes without generated
// rows, or whose centre has never been seen, contribute nothing.
template <typename T>
__global__ void __launch_bounds__(256) class_feature_match(const T* __restrict__ h, int ld, int F, const int* __restrict__ labels_r,
                                                           const int* __restrict__ labels_f, int r0, int r1, int f0, int f1,
                                                           float* __restrict__ centre, int* __restrict__ seen, float momentum,
                                                           float gscale, float* __restrict__ loss_c, T* __restrict__ dh) {
    __shared__ float red[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    int nr = 0, nf = 0;
    for (int r = r0; r < r1; ++r) nr += labels_r[r - r0] == c;
    for (int r = f0; r < f1; ++r) nf += labels_f[r - f0] == c;
    const bool was_seen = seen[c] != 0;
    float part = 0.f;
    for (int j = tid; j < F; j += 256) {
        float m = centre[(size_t)c * F + j];
        if (nr > 0) {
            float sr = 0.f;
            for (int r = r0; r < r1; ++r)
                if (labels_r[r - r0] == c) sr += to_f32(h[(size_t)r * ld + j]);
            sr /= (float)nr;
            m = was_seen ? (1.f - momentum) * m + momentum * sr : sr;
            centre[(size_t)c * F + j] = m;
        }
        if (nf > 0 && (was_seen || nr > 0)) {
            float sf = 0.f;
            for (int r = f0; r < f1; ++r)
                if (labels_f[r - f0] == c) sf += to_f32(h[(size_t)r * ld + j]);
            const float d = sf / (float)nf - m;
            part += 0.5f * d * d;
            const float g = d * gscale / (float)nf;
            for (int r = f0; r < f1; ++r)
                if (labels_f[r - f0] == c) dh[(size_t)r * ld + j] = from_f32<T>(to_f32(dh[(size_t)r * ld + j]) + g);
        }
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) {
        loss_c[c] = red[0];
        if (nr > 0) seen[c] = 1;
    }
}

// Pairwise feature reconstruction (the paper's L_G feature terms): row f0 + i (the reconstruction of image
// i) is pulled towards row r0 + i (the real image, a constant target): loss = mean_i 1/2 |h_f - h_r|^2,
// gradient (h_f - h_r) * gscale / n ADDED to row f0 + i of dh.  One block per pair.
template <typename T>
__global__ void __launch_bounds__(256) pair_feature_match(const T* __restrict__ h, int ld, int F, int r0, int f0, int n,
                                                          float gscale, float* __restrict__ loss_row, T* __restrict__ dh) {
    __shared__ float red[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    float part = 0.f;
    for (int j = tid; j < F; j += 256) {
        const float d = to_f32(h[(size_t)(f0 + i) * ld + j]) - to_f32(h[(size_t)(r0 + i) * ld + j]);
        part += 0.5f * d * d;
        T* p = dh + (size_t)(f0 + i) * ld + j;
        *p = from_f32<T>(to_f32(*p) + d * gscale / (float)n);
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) loss_row[i] = red[0];
}

__global__ void axpy_f32(float* __restrict__ y, const float* __restrict__ x, float alpha, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += alpha * x[i];
}

// eval-mode BatchNorm as one multiply-add per channel: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
__global__ void bn_eval_scale_shift(const float* __restrict__ rm, const float* __restrict__ rv,
                                    const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                    float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.f / sqrtf(rv[c] + BN_EPS));
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---- small vector helpers -----------------------------------------------------------------
__global__ void permute_f32(const float* in, float* out, int n, int d, int a, int b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[(i / d) * a + (i % d) * b];
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* in, TO* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = from_f32<TO>(to_f32(in[i]));
}

template <typename T>
__global__ void nhwc_to_nchw(const T* in, float* out, int B, int HW, int C, int ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*C*HW, hw fastest
    if (i >= (int64_t)B * C * HW) return;
    const int hw = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((int64_t)HW * C));
    out[i] = to_f32(in[((size_t)b * HW + hw) * ld + c]);
}
template <typename T>
__global__ void nchw_to_nhwc(const float* in, T* out, int B, int HW, int C, int ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*HW*ld, channel fastest
    if (i >= (int64_t)B * HW * ld) return;
    const int c = (int)(i % ld), hw = (int)((i / ld) % HW), b = (int)(i / ((int64_t)ld * HW));
    out[i] = from_f32<T>(c < C ? in[((size_t)b * C + c) * HW + hw] : 0.f);
}

// ---- stem input: NCHW f32 -> reflection-padded NHWC4 -----------------------------------------
template <typename T>
__global__ void pack_image(const float* __restrict__ x, T* __restrict__ xp, int B, int P) {
    const int Wp = P + 8, Hp = P + 6;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Hp * Wp) return;
    const int px = (int)(i % Wp), py = (int)((i / Wp) % Hp), b = (int)(i / ((int64_t)Wp * Hp));
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (px < P + 6) {
        int sy = py - 3, sx = px - 3;
        sy = sy < 0 ? -sy : (sy >= P ? 2 * P - 2 - sy : sy);
        sx = sx < 0 ? -sx : (sx >= P ? 2 * P - 2 - sx : sx);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = x[(((size_t)b * 3 + c) * P + sy) * P + sx];
    }
    store4<T>(xp + (size_t)i * 4, v);
}

// ---- loader output: uint8 HWC (channel order as stored) -> f32 NCHW in [-1, 1] ----------------
__global__ void image_u8_to_f32(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int PP) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*PP pixels
    if (i >= (int64_t)B * PP) return;
    const int b = (int)(i / PP), hw = (int)(i % PP);
#pragma unroll
    for (int c = 0; c < 3; ++c)
        out[((size_t)b * 3 + c) * PP + hw] = ((float)in[(size_t)i * 3 + c] / 255.f - 0.5f) / 0.5f;
}

// ---- label fold ------------------------------------------------------------------------------
__global__ void label_bias(const float* __restrict__ w, float* __restrict__ cbias, int cout, int y_dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;     // over y_dim*cout, n fastest
    if (i >= y_dim * cout) return;
    const int n = i % cout, c = i / cout;
    const float* p = w + ((size_t)n * (3 + y_dim) + 3 + c) * 49;
    float s = 0.f;
    for (int t = 0; t < 49; ++t) s += p[t];
    cbias[i] = s;
}
// S[b][n] = sum over the image's pixels of dy[b][pix][n]; one block per image
template <typename T>
__global__ void __launch_bounds__(256) per_image_colsum(const T* __restrict__ dy, float* __restrict__ S, int pixels,
                                                        int C) {
    const int CC = C / 8, RL = blockDim.x / CC;
    const int chunk = threadIdx.x % CC, rl = threadIdx.x / CC;
    __shared__ float sm[256][8];
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const T* base = dy + (size_t)blockIdx.x * pixels * C;
    if (rl < RL) {
        int m = rl;
        for (; m + 3 * RL < pixels; m += 4 * RL) {      // four rows in flight (one workgroup per CU), summed in order
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8<T>(base + (size_t)(m + u * RL) * C + chunk * 8, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += v[u][j];
        }
        for (; m < pixels; m += RL) {
            float v[8];
            load8<T>(base + (size_t)m * C + chunk * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sm[threadIdx.x][j] = s[j];
    __syncthreads();
    for (int col = threadIdx.x; col < C; col += blockDim.x) {
        float t = 0.f;
        for (int r = 0; r < RL; ++r) t += sm[r * CC + col / 8][col % 8];
        S[(size_t)blockIdx.x * C + col] = t;
    }
}
// Blocks [0, y_dim): one class each; its images are listed in batch order in LDS, then thread n sums
// S[b][n] over the list.  Blocks from y_dim on: the plain bias gradient, FIN_CH channels per block,
// summed over all images like any other set of partials.
__global__ void __launch_bounds__(256) label_bias_scatter(const float* __restrict__ S, const int* __restrict__ labels,
                                                          int B, int cout, int y_dim, float* __restrict__ dw,
                                                          float* __restrict__ dbias) {
    if ((int)blockIdx.x >= y_dim) {
        const int n = (blockIdx.x - y_dim) * FIN_CH + (threadIdx.x >> 6);
        double s, unused;
        sum_partials(S, B, cout, n, s, unused, false);
        if ((threadIdx.x & 63) == 0 && n < cout && dbias) dbias[n] = (float)s;
        return;
    }
    __shared__ int list[256];
    __shared__ int wcount[4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s[4] = {0.f, 0.f, 0.f, 0.f};                       // cout <= 1024
    for (int b0 = 0; b0 < B; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const bool hit = b < B && labels[b] == c;
        const unsigned long long m = __ballot(hit);
        __syncthreads();                                     // the previous list has been consumed
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) base += wcount[w]; total += wcount[w]; }
        if (hit) list[base + __popcll(m & ((1ull << lane) - 1ull))] = b;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = threadIdx.x + 256 * q;
            if (n >= cout) continue;
            for (int i = 0; i < total; ++i) s[q] += S[(size_t)list[i] * cout + n];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = threadIdx.x + 256 * q;
        if (n >= cout) continue;
        float* p = dw + ((size_t)n * (3 + y_dim) + 3 + c) * 49;
        for (int t = 0; t < 49; ++t) p[t] = s[q];
    }
}

// ---- reparameterisation + KL -----------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) reparam_kl_fwd(const T* __restrict__ h, int h_ld, const float* __restrict__ eps,
                                                       const int* __restrict__ labels, int B, int z, int y_dim,
                                                       int zin_ld, T* __restrict__ zin, float* __restrict__ z_mean,
                                                       float* __restrict__ z_sigma, float* __restrict__ kl,
                                                       float batch_size) {
    __shared__ float red[16];
    float acc = 0.f;
    // one workgroup (a fixed summation order for the KL term); four elements per thread are in flight
    // at a time, or the few dozen dependent round trips of a plain loop take 30 us
    const int n = B * z, NTH = blockDim.x;
    for (int base = threadIdx.x; base < n; base += 4 * NTH) {
        float mu[4], ls[4], ep[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NTH;
            mu[u] = ls[u] = ep[u] = 0.f;
            if (i < n) {
                const int b = i / z, j = i - b * z;
                mu[u] = to_f32(h[(size_t)b * h_ld + j]);
                ls[u] = to_f32(h[(size_t)b * h_ld + z + j]);
                ep[u] = eps[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NTH;
            if (i >= n) continue;
            const int b = i / z, j = i - b * z;
            const float sg = expf(ls[u]);
            zin[(size_t)b * zin_ld + j] = from_f32<T>(mu[u] + sg * ep[u]);
            z_mean[i] = mu[u];
            z_sigma[i] = sg;
            acc += mu[u] * mu[u] + sg * sg - 2.f * ls[u] - 1.f;     // log(sigma^2) = 2 * log_sigma
        }
    }
    // the one-hot label and the padding columns behind the latent
    const int tail = zin_ld - z;
    for (int i = threadIdx.x; i < B * tail; i += NTH) {
        const int b = i / tail, j = z + (i - b * tail);
        zin[(size_t)b * zin_ld + j] = from_f32<T>((j < z + y_dim && labels[b] == j - z) ? 1.f : 0.f);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        *kl = s / ((float)B * (float)z) / batch_size;
    }
}

template <typename T>
__global__ void reparam_kl_bwd(const T* __restrict__ dzin, int zin_ld, const float* __restrict__ eps,
                               const float* __restrict__ z_mean, const float* __restrict__ z_sigma,
                               const float* __restrict__ g_mean, const float* __restrict__ g_sigma, float kl_scale,
                               float batch_size, int B, int z, T* __restrict__ dh, int h_ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * z) return;
    const int b = i / z, j = i % z;
    const float dz = to_f32(dzin[(size_t)b * zin_ld + j]);
    const float mu = z_mean[i], sg = z_sigma[i];
    const float s = kl_scale / ((float)B * (float)z * batch_size);
    float dmu = dz + 2.f * mu * s;
    float dsg = dz * eps[i];
    if (g_mean) dmu += g_mean[i];
    if (g_sigma) dsg += g_sigma[i];
    const float dls = dsg * sg + (2.f * sg * sg - 2.f) * s;
    dh[(size_t)b * h_ld + j] = from_f32<T>(dmu);
    dh[(size_t)b * h_ld + z + j] = from_f32<T>(dls);
}

// ---- class indices: copy, keep inside [0, n_class), flag the ones that were not --------------------
__global__ void labels_prepare(const int* __restrict__ in, int* __restrict__ out, int n, int n_class, float* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v = in[i];
    if (v < 0 || v >= n_class) {
        *bad = 1.f;                     // every offender stores the same value: no ordering needed
        v = v < 0 ? 0 : n_class - 1;
    }
    out[i] = v;
}

// ---- LeakyReLU + Dropout(0.5) ------------------------------------------------------------------
template <typename T>
__global__ void lrelu_dropout_fwd(const T* __restrict__ x, const uint8_t* __restrict__ mask, T* __restrict__ y,
                                  int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    float v[8];
    load8<T>(x + i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = lrelu(v[j]);
        if (mask) v[j] = mask[i + j] ? 2.f * v[j] : 0.f;
    }
    store8<T>(y + i, v);
}
template <typename T>
__global__ void lrelu_dropout_bwd(const T* __restrict__ dy, const T* __restrict__ y, const uint8_t* __restrict__ mask,
                                  T* __restrict__ dx, int64_t n) {
    // y = lrelu(x) * (mask ? 2 : 0): sign(y) = sign(x) wherever the mask keeps the value
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i >= n) return;
    float d[8], o[8];
    load8<T>(dy + i, d);
    load8<T>(y + i, o);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float g = o[j] > 0.f ? d[j] : LRELU_SLOPE * d[j];
        if (mask) g = mask[i + j] ? 2.f * g : 0.f;
        d[j] = g;
    }
    store8<T>(dx + i, d);
}
__global__ void dropout_mask(uint8_t* mask, int64_t n, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t v = (uint64_t)i + seed * 0x9E3779B97F4A7C15ull;      // splitmix64 finaliser
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    v ^= v >> 31;
    mask[i] = (uint8_t)((v >> 40) & 1u);
}

// ---- bilinear x2 + reflection pad 1 --------------------------------------------------------------
// Up-sampled coordinate p in [0, 2h): source rows and weights exactly as
// F.interpolate(scale_factor=2, mode='bilinear', align_corners=False).
__device__ __forceinline__ void up_src(int p, int h, int& i0, int& i1, float& w0, float& w1) {
    if (p == 0) { i0 = 0; i1 = h > 1 ? 1 : 0; w0 = 1.f; w1 = 0.f; return; }
    const int i = p >> 1;
    if (p & 1) { i0 = i; i1 = i + 1 < h ? i + 1 : h - 1; w0 = 0.75f; w1 = 0.25f; }
    else { i0 = i - 1; i1 = i; w0 = 0.25f; w1 = 0.75f; }
}
__device__ __forceinline__ int unpad_reflect1(int u, int n2) {      // padded index u in [0, n2+2) -> [0, n2)
    const int p = u - 1;
    return p < 0 ? 1 : (p >= n2 ? n2 - 2 : p);
}

// Workgroups are dealt to the 8 XCDs round-robin by launch index, and each XCD has its own L2.  Rows
// that share source rows (every source row feeds 2-4 neighbouring rows here) should therefore be
// neighbours in one XCD's sequence: XCD k takes the k-th contiguous eighth of the (x, y, z) grid.
// Returns the block coordinates this workgroup works on.
__device__ __forceinline__ void xcd_block(int& bx, int& by, int& bz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, q = n >> 3, r = n & 7u;
    const unsigned v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = (int)(v % gx);
    const unsigned rest = v / gx;
    by = (int)(rest % gy);
    bz = (int)(rest / gy);
}

// grid: x over (padded column, 8-channel chunk), y = PAIR of padded rows (2k, 2k+1), z = image.  The two
// rows of a pair interpolate between the same two source rows (padded row 2k is up-sampled row 2k-1 =
// .75/.25 of rows k-1, k; padded row 2k+1 is up-sampled row 2k = .25/.75 of the same), so the four
// source vectors are loaded once per pair.
template <typename T>
__global__ void __launch_bounds__(256) upsample_pad_fwd(const T* __restrict__ x, T* __restrict__ u, int B, int h, int w,
                                                        int C) {
    const int CC = C / 8, UH = 2 * h + 2, UW = 2 * w + 2;
    // (launch order as is: this kernel is bound by its writes, and they stream best with the XCDs
    // interleaved row by row; the XCD-contiguous order of the transpose below made it 18 % slower)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= UW * CC) return;
    const int ux = t / CC, ch = t - ux * CC;
    const int uy0 = 2 * blockIdx.y, b = blockIdx.z;
    int x0, x1;
    float wx0, wx1;
    up_src(unpad_reflect1(ux, 2 * w), w, x0, x1, wx0, wx1);
    const T* base = x + (size_t)b * h * w * C + ch * 8;
    float a[8], bb[8], c[8], d[8], o[8];
    int ly0 = -1, ly1 = -1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int uy = uy0 + r;
        if (uy >= UH) break;
        int y0, y1;
        float wy0, wy1;
        up_src(unpad_reflect1(uy, 2 * h), h, y0, y1, wy0, wy1);
        if (y0 != ly0 || y1 != ly1) {             // uniform over the workgroup
            load8<T>(base + ((size_t)y0 * w + x0) * C, a);
            load8<T>(base + ((size_t)y0 * w + x1) * C, bb);
            load8<T>(base + ((size_t)y1 * w + x0) * C, c);
            load8<T>(base + ((size_t)y1 * w + x1) * C, d);
            ly0 = y0; ly1 = y1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = wy0 * (wx0 * a[j] + wx1 * bb[j]) + wy1 * (wx0 * c[j] + wx1 * d[j]);
        store8<T>(u + (((size_t)b * UH + uy) * UW + ux) * C + ch * 8, o);
    }
}

// Per-axis taps of the transpose: which padded positions feed input index i, and with what weight.
// Slots 0-3: up-sampled p = 2i-1 .. 2i+2 (padded position p+1); slot 4: padded row 0, the mirror of
// p = 1; slot 5: padded row n2+1, the mirror of p = n2-2.  Absent taps have weight 0.
__device__ __forceinline__ void up_bwd_taps(int i, int h, int (&pos)[6], float (&wt)[6]) {
    const int n2 = 2 * h;
    pos[4] = 0; wt[4] = 0.f;
    pos[5] = n2 + 1; wt[5] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = 2 * i - 1 + k;
        float wgt = 0.f;
        if (p >= 0 && p < n2) {
            int i0, i1;
            float w0, w1;
            up_src(p, h, i0, i1, w0, w1);
            wgt = (i0 == i ? w0 : 0.f) + (i1 == i ? w1 : 0.f);
        }
        pos[k] = p >= 0 && p < n2 ? p + 1 : 0;
        wt[k] = wgt;
        if (p == 1) wt[4] = wgt;
        if (p == n2 - 2) wt[5] = wgt;
    }
}

// grid: x over (column, 8-channel chunk), y = row, z = image
template <typename T>
__global__ void __launch_bounds__(256) upsample_pad_bwd(const T* __restrict__ du, T* __restrict__ dx, int B, int h, int w,
                                                        int C) {
    const int CC = C / 8, UH = 2 * h + 2, UW = 2 * w + 2;
    int bx, iy, b;
    xcd_block(bx, iy, b);
    const int t = bx * blockDim.x + threadIdx.x;
    if (t >= w * CC) return;
    const int ix = t / CC, ch = t - ix * CC;
    int py[6], px[6];
    float wy[6], wx[6];
    up_bwd_taps(iy, h, py, wy);
    up_bwd_taps(ix, w, px, wx);
    const T* base = du + (size_t)b * UH * UW * C + ch * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    // the 4 x 4 window every pixel has: all 16 loads in flight together
    float v[4][4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) load8<T>(base + ((size_t)py[a] * UW + px[c]) * C, v[a][c]);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float wgt = wy[a] * wx[c];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wgt * v[a][c][j];
        }
    // mirror taps: only the two rows / columns next to each border have them
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (a < 4 && c < 4) continue;
            const float wgt = wy[a] * wx[c];
            if (wgt == 0.f) continue;
            float e[8];
            load8<T>(base + ((size_t)py[a] * UW + px[c]) * C, e);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wgt * e[j];
        }
    store8<T>(dx + (((size_t)b * h + iy) * w + ix) * C + ch * 8, acc);
}

// ---- backward of the up-convolution at LOW resolution ------------------------------------------------------
// y = conv3x3_valid(U), U = pad(upsample(x)) is linear in x with a spatial operator A that does not mix channels:
// U[u] = sum_q A[u, q] x[q].  Therefore, with Z_t[q] = sum_p A[p + t, q] dY[p] (tap t = (ky, kx); p over the
// output pixels):      dX[q] = sum_t W_t^T Z_t[q]        dW_t = sum_q Z_t[q] (x) x[q]
// i.e. both gradients are plain GEMMs over the LOW-resolution pixels with K = 9 N: a quarter of the
// multiplications of the same gradients taken at the up-sampled resolution, and neither the gradient of U nor
// the transposed up-sampling pass exists.  This kernel builds Z [B][h][w][9][N] from dY [B][2h][2w][N].
// Per axis, x[i] appears in the padded up-sampled positions 2i .. 2i + 3 (and in the mirror rows at the borders),
// which the outputs p = u - k (k = 0..2) read: the rows 2i - 2 .. 2i + 3 of dY, with the coefficient of x[i] in U[p + k].
__device__ __forceinline__ float up_coef(int u, int i, int h) {       // coefficient of x[i] in padded up-sampled index u
    int i0, i1;
    float w0, w1;
    up_src(unpad_reflect1(u, 2 * h), h, i0, i1, w0, w1);
    return (i0 == i ? w0 : 0.f) + (i1 == i ? w1 : 0.f);
}
// 8 channels held as raw 16-byte vectors (one of bf16, two of f32) -> floats
template <typename T> __device__ __forceinline__ void unpack8(const u32x4 (&raw)[sizeof(T) / 2], float (&v)[8]);
template <> __device__ __forceinline__ void unpack8<bf16>(const u32x4 (&raw)[1], float (&v)[8]) {
    const bf16x8 a = __builtin_bit_cast(bf16x8, raw[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <> __device__ __forceinline__ void unpack8<float>(const u32x4 (&raw)[2], float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = __uint_as_float(raw[0][i]); v[4 + i] = __uint_as_float(raw[1][i]); }
}
// A workgroup takes IXB consecutive positions of one low-resolution row and NCB 8-channel chunks (IXB * NCB = 256).
// Pass 1 combines, per column of dY the row segment touches and per kernel row, the six rows of dY (one thread per
// (column, chunk): six loads); the combinations meet in LDS; pass 2 combines six columns per position and kernel
// column.  Each dY element is loaded about 1.1 times per kernel instead of nine.
// grid: x over (row segment, chunk group), y = row, z = image (of the low-resolution grid)
template <typename T>
__global__ void __launch_bounds__(256) upsample_taps_bwd(const T* __restrict__ dy, T* __restrict__ z, int B, int h, int w,
                                                         int N, int NCB) {
    extern __shared__ __attribute__((aligned(16))) float colsum[];       // [3][2 IXB + 4][NCB][8]
    const int NC = N / 8, H2 = 2 * h, W2 = 2 * w;
    const int IXB = 256 / NCB, NCOL = 2 * IXB + 4, ngrp = NC / NCB;
    int bx, iy, b;
    xcd_block(bx, iy, b);
    const int seg = bx / ngrp, cg = bx - seg * ngrp;
    const int ix0 = seg * IXB, ch0 = cg * NCB;
    float wy[6][3];
    int ry[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const int p = 2 * iy - 2 + r;
        const bool ok = p >= 0 && p < H2;
        ry[r] = ok ? p : 0;                        // rows outside the image: weight 0, any valid address
#pragma unroll
        for (int k = 0; k < 3; ++k) wy[r][k] = ok ? up_coef(p + k, iy, h) : 0.f;
    }
    const T* base = dy + (size_t)b * H2 * W2 * N;
    // (2 IXB + 4) NCB = 512 + 4 NCB <= 640 items: at most ITEMS per thread.  All of a thread's loads go out before
    // the first sum: the kernel is bound by how many bytes a CU has in flight (three workgroups fit its LDS).
    constexpr int ITEMS = 3, VW = (int)sizeof(T) / 2;           // VW 16-byte vectors per 8 channels: 1 (bf16) or 2 (f32)
    u32x4 raw[ITEMS][6][VW];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = threadIdx.x + 256 * it;
        const int col = idx / NCB, chn = idx - col * NCB;
        const int px = 2 * ix0 - 2 + col;
        const bool ok = idx < NCOL * NCB && px >= 0 && px < W2;
        const T* src = base + (size_t)(ok ? px : 0) * N + (ok ? (ch0 + chn) * 8 : 0);      // absent items: any valid address
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int q = 0; q < VW; ++q) raw[it][r][q] = *(const u32x4*)(src + (size_t)ry[r] * W2 * N + q * (16 / sizeof(T)));
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int idx = threadIdx.x + 256 * it;
        if (idx >= NCOL * NCB) break;
        const int col = idx / NCB, chn = idx - col * NCB;
        const int px = 2 * ix0 - 2 + col;
        float s[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[k][j] = 0.f;
        if (px >= 0 && px < W2) {
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float v[8];
                unpack8<T>(raw[it][r], v);
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[k][j] += wy[r][k] * v[j];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float* dst = colsum + ((size_t)(k * NCOL + col) * NCB + chn) * 8;
            *(f32x4*)dst = f32x4{s[k][0], s[k][1], s[k][2], s[k][3]};
            *(f32x4*)(dst + 4) = f32x4{s[k][4], s[k][5], s[k][6], s[k][7]};
        }
    }
    __syncthreads();
    const int ixl = threadIdx.x / NCB, chn = threadIdx.x - ixl * NCB, ix = ix0 + ixl;
    if (ix >= w) return;
    float wxk[6][3];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int px = 2 * ix - 2 + c;
        const bool okx = px >= 0 && px < W2;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) wxk[c][kx] = okx ? up_coef(px + kx, ix, w) : 0.f;
    }
    T* out = z + (((size_t)b * h + iy) * w + ix) * (size_t)(9 * N) + (ch0 + chn) * 8;
    // one kernel row at a time: 24 accumulators live instead of 72, so that three workgroups fit a CU's registers
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float acc[3][8];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[kx][j] = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float* src = colsum + ((size_t)(k * NCOL + 2 * ixl + c) * NCB + chn) * 8;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[kx][j] += wxk[c][kx] * lo[j];
                    acc[kx][4 + j] += wxk[c][kx] * hi[j];
                }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) store8<T>(out + (k * 3 + kx) * N, acc[kx]);
    }
}

// ---- forward of the up-convolution at LOW resolution --------------------------------------------------------
// By the same linearity, y[p] = bias + sum_t sum_q A[p + t, q] V_t[q] with V_t[q] = W_t x[q]: the channel mixing is
// ONE GEMM over the low-resolution pixels with 9 N output columns (a quarter of the multiplications of the
// convolution over the up-sampled tensor, which never exists), and this kernel applies the spatial operator:
// v [B][h][w][9][N] -> y [B][2h][2w][N].  A thread owns one low-resolution position and 8 channels, i.e. the 2 x 2
// output pixels 2a + {0, 1}, 2b + {0, 1}; they read taps of the 3 x 3 positions around (a, b) only.
// grid: x over (column, 8-channel chunk), y = row, z = image (of the low-resolution grid)
template <typename T>
__global__ void __launch_bounds__(256) upsample_taps_fwd(const T* __restrict__ v, const float* __restrict__ bias,
                                                         T* __restrict__ y, int B, int h, int w, int N) {
    const int NC = N / 8;
    int bx, a, b;
    xcd_block(bx, a, b);
    const int t = bx * blockDim.x + threadIdx.x;
    if (t >= w * NC) return;
    const int ix = t / NC, ch = t - ix * NC;
    // coefficient of x[a - 1 + d] in the padded up-sampled row that output row 2a + par reads at kernel row k
    float wy[2][3][3], wx[2][3][3];
#pragma unroll
    for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int qy = a - 1 + d, qx = ix - 1 + d;
                wy[par][k][d] = (qy >= 0 && qy < h) ? up_coef(2 * a + par + k, qy, h) : 0.f;
                wx[par][k][d] = (qx >= 0 && qx < w) ? up_coef(2 * ix + par + k, qx, w) : 0.f;
            }
    float acc[2][2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float bv = bias ? bias[ch * 8 + j] : 0.f;
        acc[0][0][j] = acc[0][1][j] = acc[1][0][j] = acc[1][1][j] = bv;
    }
    const T* base = v + (size_t)b * h * w * (9 * N) + ch * 8;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int qy = min(max(a - 1 + dy, 0), h - 1);         // positions outside the image: weight 0, any valid address
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int qx = min(max(ix - 1 + dx, 0), w - 1);
            const T* src = base + ((size_t)qy * w + qx) * (9 * N);
            float val[9][8];
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) load8<T>(src + tp * N, val[tp]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int py = 0; py < 2; ++py)
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const float c = wy[py][ky][dy] * wx[px][kx][dx];
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[py][px][j] += c * val[ky * 3 + kx][j];
                        }
        }
    }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px)
            store8<T>(y + (((size_t)b * 2 * h + 2 * a + py) * (2 * w) + 2 * ix + px) * N + ch * 8, acc[py][px]);
}

// transpose of ReflectionPad2d(pad)
__device__ __forceinline__ int fold_list(int y, int H, int pad, int (&pos)[3]) {
    int n = 0;
    pos[n++] = y + pad;
    if (y >= 1 && y <= pad) pos[n++] = pad - y;
    if (y <= H - 2 && y >= H - 1 - pad) pos[n++] = pad + 2 * H - 2 - y;
    return n;
}
template <typename T>
__global__ void reflect_fold(const T* __restrict__ dxp, T* __restrict__ dx, int B, int H, int W, int C, int pad) {
    const int CC = C / 8, Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H * W * CC) return;
    const int ch = (int)(i % CC);
    const int x = (int)((i / CC) % W), y = (int)((i / ((int64_t)CC * W)) % H);
    const int b = (int)(i / ((int64_t)CC * W * H));
    int py[3], px[3];
    const int ny = fold_list(y, H, pad, py), nx = fold_list(x, W, pad, px);
    const T* base = dxp + (size_t)b * Hp * Wp * C + ch * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int a = 0; a < ny; ++a)
        for (int c = 0; c < nx; ++c) {
            float v[8];
            load8<T>(base + ((size_t)py[a] * Wp + px[c]) * C, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
    store8<T>(dx + (size_t)i * 8, acc);
}

// ---- tanh + MSE -------------------------------------------------------------------------------------
template <typename T>
__global__ void tanh_fwd(const T* __restrict__ pre, float* __restrict__ dec, int B, int PP) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*PP pixels
    if (i >= (int64_t)B * PP) return;
    const int b = (int)(i / PP), hw = (int)(i % PP);
    float v[4];
    load4<T>(pre + (size_t)i * 4, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) dec[((size_t)b * 3 + c) * PP + hw] = tanhf(v[c]);
}
// Tail finished from per-column partial sums: v[b][y][x'][kw*4 + n] holds the 7 row taps of column
// x' for kernel column kw, so pre[b][y][x][n] = bias[n] + sum_kw v[b][y][x + kw][kw*4 + n].
template <typename T>
__global__ void tail_sum_tanh(const T* __restrict__ v, const float* __restrict__ bias, T* __restrict__ pre,
                              float* __restrict__ dec, int B, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*P*P pixels
    const int PP = P * P;
    if (i >= (int64_t)B * PP) return;
    const int b = (int)(i / PP), hw = (int)(i % PP);
    const int y = hw / P, x = hw - y * P;
    const T* row = v + (((size_t)b * P + y) * (P + 6) + x) * 32;
    float s[4] = {bias[0], bias[1], bias[2], 0.f};
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
        float t[4];
        load4<T>(row + kw * 32 + kw * 4, t);
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] += t[c];
    }
    store4<T>(pre + (size_t)i * 4, s);
#pragma unroll
    for (int c = 0; c < 3; ++c) dec[((size_t)b * 3 + c) * PP + hw] = tanhf(s[c]);
}
// Input gradient of a 7x7 convolution with 3 (stored as 4) input channels from per-column partials: v[b][y][j][kw*4 + c]
// holds, for input column j of the incoming gradient and kernel column kw, the sum over the 7 kernel rows and all
// output channels; out[b][y][x][c] = sum_kw v[b][y][x - kw][kw*4 + c] over the columns that exist (0 <= x - kw < P),
// x in [0, P + 6): the gradient with respect to the padded image, as 8-channel pixels (channels 3.. zero).
template <typename T>
__global__ void window_col_sum(const T* __restrict__ v, T* __restrict__ out, int B, int rows, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B*rows*(P+6) pixels
    const int W = P + 6;
    if (i >= (int64_t)B * rows * W) return;
    const int x = (int)(i % W);
    const int64_t by = i / W;                                            // b * rows + y
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
        const int j = x - kw;
        if (j < 0 || j >= P) continue;
        float t[4];
        load4<T>(v + ((size_t)by * P + j) * 32 + kw * 4, t);
#pragma unroll
        for (int c = 0; c < 3; ++c) s[c] += t[c];
    }
    store8<T>(out + (size_t)i * 8, s);
}
template <typename T>
__global__ void tanh_bwd(const float* __restrict__ d_dec, const float* __restrict__ dec, T* __restrict__ dpre,
                         T* __restrict__ win, int B, int PP, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * PP) return;
    const int b = (int)(i / PP), hw = (int)(i % PP);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t o = ((size_t)b * 3 + c) * PP + hw;
        const float t = dec[o];
        v[c] = d_dec[o] * (1.f - t * t);
    }
    store8<T>(dpre + (size_t)i * 8, v);
    if (win) {
        const int y = hw / P, x = hw - y * P;
        const float v4[4] = {v[0], v[1], v[2], 0.f};
        store4<T>(win + (((size_t)b * (P + 12) + y + 6) * (P + 14) + x + 6) * 4, v4);
    }
}

__global__ void __launch_bounds__(256) mse_partial(const float* __restrict__ dec, const float* __restrict__ x,
                                                   float* __restrict__ d_dec, float* __restrict__ ws, int64_t n,
                                                   float coef) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = dec[i] - x[i];
        acc += d * d;
        d_dec[i] = coef * d;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256) mse_finalize(const float* ws, int nblk, float* ll, float scale) {
    __shared__ double red[4];
    double a = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) a += ws[b];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) *ll = (float)((red[0] + red[1] + red[2] + red[3]) * scale);
}

// ---- AdamW ------------------------------------------------------------------------------------------
// One element of torch.optim.AdamW's update, with the scalars rounded where torch rounds them: every
// hyper-parameter expression (1 - lr * wd, 1 - beta, lr / bias_correction1, sqrt(bias_correction2)) is
// evaluated in double on the host (or by adamw_tick) and enters the f32 arithmetic as one float.
struct AdamScalars {
    float decay;          // 1 - lr * weight_decay
    float w1;             // 1 - beta1 (lerp weight of the first moment)
    float b2, omb2;       // beta2, 1 - beta2
    float eps;
};
__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, const AdamScalars& k, float step_size,
                                           float bc2_sqrt) {
    p *= k.decay;
    const float d = g - m;                                    // exp_avg.lerp_(grad, 1 - beta1)
    m = k.w1 < 0.5f ? m + k.w1 * d : g - d * (1.f - k.w1);
    v = v * k.b2 + k.omb2 * (g * g);                          // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + k.eps;
    p -= step_size * (m / denom);                             // addcdiv_(exp_avg, denom, -step_size)
}
__device__ __forceinline__ void adamw_span(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n, const AdamScalars& k, float step_size,
                                           float bc2_sqrt) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 pp = *(f32x4*)(p + i), gg = *(const f32x4*)(g + i), mm = *(f32x4*)(m + i), vv = *(f32x4*)(v + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = pp[j], b = mm[j], c = vv[j];
            adamw_elem(a, gg[j], b, c, k, step_size, bc2_sqrt);
            pp[j] = a; mm[j] = b; vv[j] = c;
        }
        *(f32x4*)(p + i) = pp; *(f32x4*)(m + i) = mm; *(f32x4*)(v + i) = vv;
    } else {
        for (int64_t q = i; q < n; ++q) adamw_elem(p[q], g[q], m[q], v[q], k, step_size, bc2_sqrt);
    }
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, int64_t n, const AdamScalars k, float step_size, float bc2_sqrt) {
    adamw_span(p, g, m, v, n, k, step_size, bc2_sqrt);
}

// Device-resident step counter: the bias corrections are recomputed on the device so that a
// captured HIP graph of the step advances AdamW exactly like eager launches do.
// bc[0] = lr / (1 - beta1^t), bc[1] = sqrt(1 - beta2^t).
__global__ void adamw_tick(long long* step, float* bc, double lr, double b1, double b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long t = *step + 1;
        *step = t;
        bc[0] = (float)(lr / (1.0 - pow(b1, (double)t)));
        bc[1] = (float)sqrt(1.0 - pow(b2, (double)t));
    }
}
__global__ void adamw_kernel_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, const AdamScalars k, const float* __restrict__ bc) {
    adamw_span(p, g, m, v, n, k, bc[0], bc[1]);
}
__global__ void dropout_mask_dev(uint8_t* mask, int64_t n, const long long* seed_ctr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t v = (uint64_t)i + (uint64_t)(*seed_ctr) * 0x9E3779B97F4A7C15ull;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    v ^= v >> 31;
    mask[i] = (uint8_t)((v >> 40) & 1u);
}
__global__ void counter_tick(long long* c) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += 1;
}

// ---- launch helpers ------------------------------------------------------------------------------------
int col_block_of(int C) { return C <= 1024 ? C : 512; }
int red_threads(int C) { const int CC = C / 8; return CC * (256 / CC); }
int red_blocks(int M, int C) {
    const int RL = 256 / (C / 8);
    int nb = ceil_div(M, (int64_t)RL * 8);
    return nb < 1 ? 1 : (nb > 512 ? 512 : nb);
}
// Workspace of a reduction over rows[M][C], in floats: CVG_WS_HEADER arrival counters ([column blocks][ngroups + 1],
// at the start so that calls with different shapes that share one workspace never write data over them), block
// partials [2][nblk][C], level-2 partials (doubles) [2][ngroups][C].
int red_groups(int nblk) { return ceil_div(nblk, FIN_GROUP); }
int red_counters(int C, int nblk) { return ceil_div(C, col_block_of(C)) * (red_groups(nblk) + 1); }
int red_rows(int M, int C) {                        // rows R such that 2 * R * C floats hold all of it
    const int nblk = red_blocks(M, col_block_of(C)), ng = red_groups(nblk);
    return nblk + 2 * ng + ceil_div(CVG_WS_HEADER, 2 * C);
}
// Points `a` at a workspace of red_rows(M, C) rows starting at ws and arms the in-launch totals.  Shapes with more
// column blocks than the header has counters for (C > 600000) keep fin == FIN_NONE: the caller then launches the
// separate finalize kernel.
int red_finalize_setup(RedArgs& a, float* ws, int fin, const char* who) {
    CVG_CHECK(((uintptr_t)ws & 7) == 0, "%s: the workspace must be 8-byte aligned", who);
    a.ws = ws + CVG_WS_HEADER;
    a.fin = (red_counters(a.C, a.nblk) <= CVG_WS_HEADER && !cvg_switches().split_finalize) ? fin : FIN_NONE;
    a.ws2 = (double*)(a.ws + (size_t)2 * a.nblk * a.C);
    a.tickets = (unsigned*)ws;
    return 0;
}
int check_rows(const char* who, int M, int C) {
    CVG_CHECK(M > 0 && C >= 8 && C % 8 == 0 && C <= 65535 * 512, "%s: M %d C %d (C must be a positive multiple of 8)", who, M, C);
    return 0;
}
int col_block(int C) { return col_block_of(C); }
int grid1(int64_t n, int per = 256) { return ceil_div(n, per); }

}  // namespace

#define DISPATCH(dtype, ...) do { if ((dtype) == CVG_BF16) { typedef bf16 T; __VA_ARGS__; } \
    else if ((dtype) == CVG_F32) { typedef float T; __VA_ARGS__; } else CVG_FAIL("bad dtype %d", (int)(dtype)); } while (0)

extern "C" {

int cvg_pack_image(const float* x, void* xpad, int32_t B, int32_t P, int32_t dtype, cvg_stream stream) {
    CVG_CHECK(x && xpad && B > 0 && P >= 8, "pack_image: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (int64_t)B * (P + 6) * (P + 8);
    DISPATCH(dtype, pack_image<T><<<grid1(n), 256, 0, s>>>(x, (T*)xpad, B, P));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_image_u8_to_f32(const uint8_t* hwc, float* nchw, int32_t B, int32_t P, cvg_stream stream) {
    CVG_CHECK(hwc && nchw && B > 0 && P > 0, "image_u8_to_f32: bad args");
    image_u8_to_f32<<<grid1((int64_t)B * P * P), 256, 0, (hipStream_t)stream>>>(hwc, nchw, B, P * P);
    CVG_LAUNCH_CHECK();
    return 0;
}

int32_t cvg_bn_stats_blocks(int32_t M, int32_t C) { return red_rows(M, C); }

int cvg_bn_stats(const void* x, int32_t dtype, int32_t M, int32_t C, float* stats, float* running_mean,
                 float* running_var, int64_t* nbt, float* ws, cvg_stream stream) {
    if (check_rows("bn_stats", M, C)) return -1;
    CVG_CHECK(x && stats && ws, "bn_stats: null");
    hipStream_t s = (hipStream_t)stream;
    const int Cb = col_block(C);
    const bool raw_sums = cvg_switches().bn_raw_sums;       // A/B switch: sums about 0
    float* shift = raw_sums ? nullptr : stats + C;      // parked in the slot the finalize overwrites last
    RedArgs a = {x, nullptr, nullptr, nullptr, nullptr, nullptr, ws, shift, M, C, C, red_blocks(M, Cb), Cb};
    if (red_finalize_setup(a, ws, FIN_STATS, "bn_stats")) return -1;
    a.stats_out = stats; a.running_mean = running_mean; a.running_var = running_var; a.nbt = (long long*)nbt;
    DISPATCH(dtype, colreduce<T, RED_STATS><<<dim3(a.nblk, ceil_div(C, Cb)), red_threads(Cb), 0, s>>>(a));
    CVG_LAUNCH_CHECK();
    if (a.fin == FIN_NONE) {
        stats_finalize<<<grid1(C, FIN_CH), 256, 0, s>>>(a.ws, a.nblk, M, C, stats, running_mean, running_var, (long long*)nbt, shift);
        CVG_LAUNCH_CHECK();
    }
    return 0;
}

int cvg_bn_stats_finalize(const float* partials, int32_t nblk, int32_t M, int32_t C, float* stats, float* running_mean,
                          float* running_var, int64_t* nbt, cvg_stream stream) {
    CVG_CHECK(partials && stats && nblk > 0 && M > 0 && C > 0, "bn_stats_finalize: bad args");
    stats_finalize<<<grid1(C, FIN_CH), 256, 0, (hipStream_t)stream>>>(partials, nblk, M, C, stats, running_mean, running_var,
                                                                      (long long*)nbt, nullptr);
    CVG_LAUNCH_CHECK();
    return 0;
}

// ---- cvg_bn_stats without a workspace argument: the partials live in a static device array ----------
constexpr int STATIC_MAX_C = 1024, STATIC_MAX_ROWS = 576;      // 512 block rows + 2 * 32 rows of level-2 doubles
static __device__ __align__(256) float scratch[CVG_WS_HEADER + 2 * STATIC_MAX_ROWS * STATIC_MAX_C];     // 4.7 MiB per GPU

// The array's device address, looked up once per device with that device current.  In host code the
// name `scratch` is only a handle for this lookup: every launch below takes the returned pointer.
static int static_scratch(float** out) {
    constexpr int MAX_DEV = 64;
    static float* ptr[MAX_DEV];
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    CVG_CHECK(e == hipSuccess && dev >= 0 && dev < MAX_DEV, "bn_stats_static: no current device (%s)", hipGetErrorString(e));
    if (!ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        CVG_CHECK(e == hipSuccess && p, "bn_stats_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
        ptr[dev] = p;
    }
    *out = ptr[dev];
    return 0;
}

int cvg_bn_stats_static(const void* x, int32_t dtype, int32_t M, int32_t C, float* stats, float* running_mean,
                        float* running_var, int64_t* nbt, cvg_stream stream) {
    if (check_rows("bn_stats_static", M, C)) return -1;
    CVG_CHECK(C <= STATIC_MAX_C, "bn_stats_static: C %d exceeds the static scratch (at most %d channels); use cvg_bn_stats "
              "with a caller-owned workspace", C, STATIC_MAX_C);
    CVG_CHECK(x && stats, "bn_stats_static: null");
    float* ws = nullptr;
    if (static_scratch(&ws)) return -1;
    hipStream_t s = (hipStream_t)stream;
    // Every partial that is read was written by this launch first, so nothing left in the array by an earlier
    // call is ever read; the arrival counters at the start of the array are zero in the library's image and
    // every launch leaves them at zero: no clearing needed.
    CVG_CHECK((size_t)2 * red_rows(M, C) * C <= sizeof(scratch) / sizeof(float) + 2 * C,
              "bn_stats_static: internal: workspace of %d rows", red_rows(M, C));
    RedArgs a = {x, nullptr, nullptr, nullptr, nullptr, nullptr, ws, stats + C, M, C, C, red_blocks(M, C), C};
    if (red_finalize_setup(a, ws, FIN_STATS, "bn_stats_static")) return -1;
    a.stats_out = stats; a.running_mean = running_mean; a.running_var = running_var; a.nbt = (long long*)nbt;
    DISPATCH(dtype, colreduce<T, RED_STATS><<<dim3(a.nblk, 1), red_threads(C), 0, s>>>(a));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_bn_eval_stats(const float* rm, const float* rv, int32_t C, float* stats, cvg_stream stream) {
    CVG_CHECK(rm && rv && stats && C > 0, "bn_eval_stats: bad args");
    eval_stats<<<grid1(C, 128), 128, 0, (hipStream_t)stream>>>(rm, rv, C, stats);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_bn_eval_scale_shift(const float* rm, const float* rv, const float* gamma, const float* beta, int32_t C,
                            float* scale, float* shift, cvg_stream stream) {
    CVG_CHECK(rm && rv && gamma && beta && scale && shift && C > 0, "bn_eval_scale_shift: bad args");
    bn_eval_scale_shift<<<grid1(C), 256, 0, (hipStream_t)stream>>>(rm, rv, gamma, beta, C, scale, shift);
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_bn_act_fwd(const void* x, const void* res, void* y, int32_t dtype, int32_t M, int32_t C, const float* stats,
                   const float* gamma, const float* beta, cvg_stream stream) {
    if (check_rows("bn_act_fwd", M, C)) return -1;
    CVG_CHECK(x && y && stats && gamma && beta, "bn_act_fwd: null");
    const int Cb = col_block(C), RL = 256 / (Cb / 8);
    int nb = ceil_div(M, (int64_t)RL * 4);
    if (nb > 4096) nb = 4096;
    DISPATCH(dtype, bn_act_fwd<T><<<dim3(nb, ceil_div(C, Cb)), red_threads(Cb), 0, (hipStream_t)stream>>>(
                        (const T*)x, (const T*)res, (T*)y, M, C, Cb, stats, gamma, beta));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_bn_act_bwd(const void* dy, const void* y, const void* x, void* dx, int32_t dtype, int32_t M, int32_t C,
                   const float* stats, const float* gamma, const float* beta, float* dgamma, float* dbeta, float* ws,
                   cvg_stream stream) {
    if (check_rows("bn_act_bwd", M, C)) return -1;
    CVG_CHECK(dy && x && dx && stats && gamma && dgamma && dbeta && ws, "bn_act_bwd: null");
    hipStream_t s = (hipStream_t)stream;
    const int Cb = col_block(C);
    RedArgs a = {x, dy, y, stats, gamma, beta, ws, nullptr, M, C, C, red_blocks(M, Cb), Cb};
    const int gate = y ? 0 : (beta ? 1 : 2);     // no y and no beta: dy is already gated
    if (red_finalize_setup(a, ws, FIN_SUMS, "bn_act_bwd")) return -1;
    a.out0 = dbeta; a.out1 = dgamma; a.md = 1; a.ma = 1; a.mb = 0;
    const dim3 rgrid(a.nblk, ceil_div(C, Cb));
    if (gate == 0) DISPATCH(dtype, colreduce<T, RED_BNBWD><<<rgrid, red_threads(Cb), 0, s>>>(a));
    else if (gate == 1) DISPATCH(dtype, colreduce<T, RED_BNBWD_X><<<rgrid, red_threads(Cb), 0, s>>>(a));
    else DISPATCH(dtype, colreduce<T, RED_BNBWD_PLAIN><<<rgrid, red_threads(Cb), 0, s>>>(a));
    CVG_LAUNCH_CHECK();
    if (a.fin == FIN_NONE) {
        sum_finalize<<<grid1(C, FIN_CH), 256, 0, s>>>(a.ws, a.nblk, C, dbeta, dgamma, 1, 1, 0);
        CVG_LAUNCH_CHECK();
    }
    const int RL = 256 / (Cb / 8);
    int nb = ceil_div(M, (int64_t)RL * 4);
    if (nb > 4096) nb = 4096;
#define BN_DX(G) DISPATCH(dtype, bn_act_bwd_dx<T, G><<<dim3(nb, ceil_div(C, Cb)), red_threads(Cb), 0, s>>>( \
                              (const T*)dy, (const T*)y, (const T*)x, (T*)dx, M, C, Cb, stats, gamma, beta, dgamma, dbeta))
    if (gate == 0) BN_DX(0);
    else if (gate == 1) BN_DX(1);
    else BN_DX(2);
#undef BN_DX
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_colsum(const void* x, int32_t dtype, int32_t M, int32_t C, int32_t ld, float* out, int32_t md, int32_t ma,
               int32_t mb, float* ws, cvg_stream stream) {
    CVG_CHECK(M > 0 && C >= 8 && C % 8 == 0 && C <= 65535 * 512, "colsum: C %d must be a positive multiple of 8", C);
    CVG_CHECK(x && out && ws && ld >= C && ld % 8 == 0 && md > 0, "colsum: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int Cb = col_block(C);
    RedArgs a = {x, nullptr, nullptr, nullptr, nullptr, nullptr, ws, nullptr, M, C, ld, red_blocks(M, Cb), Cb};
    if (red_finalize_setup(a, ws, FIN_SUMS, "colsum")) return -1;
    a.out0 = out; a.out1 = nullptr; a.md = md; a.ma = ma; a.mb = mb;
    DISPATCH(dtype, colreduce<T, RED_SUM><<<dim3(a.nblk, ceil_div(C, Cb)), red_threads(Cb), 0, s>>>(a));
    CVG_LAUNCH_CHECK();
    if (a.fin == FIN_NONE) {
        sum_finalize<<<grid1(C, FIN_CH), 256, 0, s>>>(a.ws, a.nblk, C, out, nullptr, md, ma, mb);
        CVG_LAUNCH_CHECK();
    }
    return 0;
}

int cvg_permute_f32(const float* in, float* out, int32_t n, int32_t d, int32_t a, int32_t b, cvg_stream stream) {
    CVG_CHECK(in && out && n > 0 && d > 0, "permute: bad args");
    permute_f32<<<grid1(n), 256, 0, (hipStream_t)stream>>>(in, out, n, d, a, b);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_label_bias(const float* w, float* cbias, int32_t cout, int32_t y_dim, cvg_stream stream) {
    CVG_CHECK(w && cbias && cout > 0 && y_dim > 0, "label_bias: bad args");
    label_bias<<<grid1((int64_t)cout * y_dim), 256, 0, (hipStream_t)stream>>>(w, cbias, cout, y_dim);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_label_bias_bwd(const void* dy, int32_t dtype, const int32_t* labels, int32_t B, int32_t pixels, int32_t cout,
                       int32_t y_dim, float* dw, float* dbias, float* ws, cvg_stream stream) {
    if (check_rows("label_bias_bwd", pixels, cout)) return -1;
    CVG_CHECK(cout <= 1024, "label_bias_bwd: at most 1024 stem channels (got %d)", cout);
    CVG_CHECK(dy && labels && dw && ws && B > 0, "label_bias_bwd: null");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype, per_image_colsum<T><<<B, red_threads(cout), 0, s>>>((const T*)dy, ws, pixels, cout));
    CVG_LAUNCH_CHECK();
    label_bias_scatter<<<y_dim + grid1(cout, FIN_CH), 256, 0, s>>>(ws, labels, B, cout, y_dim, dw, dbias);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_reparam_kl_fwd(const void* h, int32_t dtype, int32_t h_ld, const float* eps, const int32_t* labels, int32_t B,
                       int32_t z, int32_t y_dim, int32_t zin_ld, void* zin, float* z_mean, float* z_sigma, float* kl,
                       float batch_size, cvg_stream stream) {
    CVG_CHECK(h && eps && labels && zin && z_mean && z_sigma && kl, "reparam_kl_fwd: null");
    CVG_CHECK(zin_ld >= z + y_dim && zin_ld % 8 == 0 && B > 0 && z > 0 && h_ld >= 2 * z, "reparam_kl_fwd: bad dims");
    DISPATCH(dtype, reparam_kl_fwd<T><<<1, 1024, 0, (hipStream_t)stream>>>((const T*)h, h_ld, eps, labels, B, z, y_dim, zin_ld,
                                                                          (T*)zin, z_mean, z_sigma, kl, batch_size));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_reparam_kl_bwd(const void* dzin, int32_t dtype, int32_t zin_ld, const float* eps, const float* z_mean,
                       const float* z_sigma, const float* g_mean, const float* g_sigma, float kl_scale,
                       float batch_size, int32_t B, int32_t z, void* dh, int32_t h_ld, cvg_stream stream) {
    CVG_CHECK(dzin && eps && z_mean && z_sigma && dh && B > 0 && z > 0 && h_ld >= 2 * z, "reparam_kl_bwd: bad args");
    DISPATCH(dtype, reparam_kl_bwd<T><<<grid1((int64_t)B * z), 256, 0, (hipStream_t)stream>>>(
                        (const T*)dzin, zin_ld, eps, z_mean, z_sigma, g_mean, g_sigma, kl_scale, batch_size, B, z, (T*)dh, h_ld));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_labels_prepare(const int32_t* labels, int32_t* out, int32_t n, int32_t n_class, float* bad, cvg_stream stream) {
    CVG_CHECK(labels && out && bad && n > 0 && n_class > 0, "labels_prepare: bad args");
    labels_prepare<<<grid1(n), 256, 0, (hipStream_t)stream>>>(labels, out, n, n_class, bad);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_lrelu_dropout_fwd(const void* x, const uint8_t* mask, void* y, int32_t dtype, int64_t n, cvg_stream stream) {
    CVG_CHECK(x && y && n > 0 && n % 8 == 0, "lrelu_dropout_fwd: n must be a positive multiple of 8");
    DISPATCH(dtype, lrelu_dropout_fwd<T><<<grid1(n / 8), 256, 0, (hipStream_t)stream>>>((const T*)x, mask, (T*)y, n));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_lrelu_dropout_bwd(const void* dy, const void* y, const uint8_t* mask, void* dx, int32_t dtype, int64_t n,
                          cvg_stream stream) {
    CVG_CHECK(dy && y && dx && n > 0 && n % 8 == 0, "lrelu_dropout_bwd: n must be a positive multiple of 8");
    DISPATCH(dtype, lrelu_dropout_bwd<T><<<grid1(n / 8), 256, 0, (hipStream_t)stream>>>((const T*)dy, (const T*)y, mask,
                                                                                      (T*)dx, n));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_dropout_mask(uint8_t* mask, int64_t n, uint64_t seed, cvg_stream stream) {
    CVG_CHECK(mask && n > 0, "dropout_mask: bad args");
    dropout_mask<<<grid1(n), 256, 0, (hipStream_t)stream>>>(mask, n, seed);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_upsample_pad_fwd(const void* x, void* u, int32_t dtype, int32_t B, int32_t h, int32_t w, int32_t C,
                         cvg_stream stream) {
    CVG_CHECK(x && u && B > 0 && h > 0 && w > 0 && C % 8 == 0, "upsample_pad_fwd: bad args");
    CVG_CHECK(2 * h + 2 <= 65535 && B <= 65535, "upsample_pad_fwd: grid too large");
    const dim3 grid(grid1((int64_t)(2 * w + 2) * (C / 8)), h + 1, B);
    DISPATCH(dtype, upsample_pad_fwd<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)x, (T*)u, B, h, w, C));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_upsample_pad_bwd(const void* du, void* dx, int32_t dtype, int32_t B, int32_t h, int32_t w, int32_t C,
                         cvg_stream stream) {
    CVG_CHECK(du && dx && B > 0 && h > 0 && w > 0 && C % 8 == 0, "upsample_pad_bwd: bad args");
    CVG_CHECK(h <= 65535 && B <= 65535, "upsample_pad_bwd: grid too large");
    const dim3 grid(grid1((int64_t)w * (C / 8)), h, B);
    DISPATCH(dtype, upsample_pad_bwd<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)du, (T*)dx, B, h, w, C));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_upsample_taps_fwd(const void* v, const float* bias, void* y, int32_t dtype, int32_t B, int32_t h, int32_t w,
                          int32_t N, cvg_stream stream) {
    CVG_CHECK(v && y && B > 0 && h > 0 && w > 0 && N > 0 && N % 8 == 0, "upsample_taps_fwd: bad args");
    CVG_CHECK(h <= 65535 && B <= 65535, "upsample_taps_fwd: grid too large");
    const dim3 grid(grid1((int64_t)w * (N / 8)), h, B);
    DISPATCH(dtype, upsample_taps_fwd<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)v, bias, (T*)y, B, h, w, N));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_upsample_taps_bwd(const void* dy, void* z, int32_t dtype, int32_t B, int32_t h, int32_t w, int32_t N,
                          cvg_stream stream) {
    CVG_CHECK(dy && z && B > 0 && h > 0 && w > 0 && N > 0 && N % 8 == 0, "upsample_taps_bwd: bad args");
    CVG_CHECK(h <= 65535 && B <= 65535, "upsample_taps_bwd: grid too large");
    // 8-channel chunks per workgroup: the largest power of two <= 32 that divides N / 8 (256 threads = positions x chunks)
    int ncb = 1;
    while (ncb < 32 && (N / 8) % (ncb * 2) == 0) ncb *= 2;
    const int ixb = 256 / ncb;
    const size_t lds = (size_t)3 * (2 * ixb + 4) * ncb * 8 * sizeof(float);
    CVG_CHECK(lds <= 64 * 1024, "upsample_taps_bwd: internal: %zu bytes of LDS", lds);
    const dim3 grid(ceil_div(w, ixb) * ((N / 8) / ncb), h, B);
    DISPATCH(dtype, upsample_taps_bwd<T><<<grid, 256, lds, (hipStream_t)stream>>>((const T*)dy, (T*)z, B, h, w, N, ncb));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_reflect_fold(const void* dxp, void* dx, int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, int32_t pad,
                     cvg_stream stream) {
    CVG_CHECK(dxp && dx && B > 0 && pad >= 0 && pad < H && pad < W && C % 8 == 0, "reflect_fold: bad args");
    const int64_t n = (int64_t)B * H * W * (C / 8);
    DISPATCH(dtype, reflect_fold<T><<<grid1(n), 256, 0, (hipStream_t)stream>>>((const T*)dxp, (T*)dx, B, H, W, C, pad));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_tanh_fwd(const void* pre, float* dec, int32_t dtype, int32_t B, int32_t P, cvg_stream stream) {
    CVG_CHECK(pre && dec && B > 0 && P > 0, "tanh_fwd: bad args");
    DISPATCH(dtype, tanh_fwd<T><<<grid1((int64_t)B * P * P), 256, 0, (hipStream_t)stream>>>((const T*)pre, dec, B, P * P));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_tail_sum_tanh(const void* v, const float* bias, void* pre, float* dec, int32_t dtype, int32_t B, int32_t P,
                      cvg_stream stream) {
    CVG_CHECK(v && bias && pre && dec && B > 0 && P > 0, "tail_sum_tanh: bad args");
    DISPATCH(dtype, tail_sum_tanh<T><<<grid1((int64_t)B * P * P), 256, 0, (hipStream_t)stream>>>((const T*)v, bias, (T*)pre,
                                                                                                dec, B, P));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_window_col_sum(const void* v, void* out, int32_t dtype, int32_t B, int32_t rows, int32_t P, cvg_stream stream) {
    CVG_CHECK(v && out && B > 0 && rows > 0 && P > 0, "window_col_sum: bad args");
    DISPATCH(dtype, window_col_sum<T><<<grid1((int64_t)B * rows * (P + 6)), 256, 0, (hipStream_t)stream>>>((const T*)v, (T*)out,
                                                                                                        B, rows, P));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_tanh_bwd(const float* d_dec, const float* dec, void* dpre, void* dpre_win, int32_t dtype, int32_t B, int32_t P,
                 cvg_stream stream) {
    CVG_CHECK(d_dec && dec && dpre && B > 0 && P > 0, "tanh_bwd: bad args");
    DISPATCH(dtype, tanh_bwd<T><<<grid1((int64_t)B * P * P), 256, 0, (hipStream_t)stream>>>(d_dec, dec, (T*)dpre,
                                                                                           (T*)dpre_win, B, P * P, P));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_mse_fwd_bwd(const float* dec, const float* x, float* d_dec, float* ll, int64_t numel, float batch_size,
                    float scale, float* ws, cvg_stream stream) {
    CVG_CHECK(dec && x && d_dec && ll && ws && numel > 0 && batch_size > 0, "mse: bad args");
    hipStream_t s = (hipStream_t)stream;
    int nb = grid1(numel, 256 * 8);
    if (nb > 1024) nb = 1024;
    const float coef = scale * 100.f / ((float)numel * batch_size);
    mse_partial<<<nb, 256, 0, s>>>(dec, x, d_dec, ws, numel, coef);
    CVG_LAUNCH_CHECK();
    mse_finalize<<<1, 256, 0, s>>>(ws, nb, ll, 50.f / ((float)numel * batch_size));
    CVG_LAUNCH_CHECK();
    return 0;
}

static AdamScalars adam_scalars(double lr, double beta1, double beta2, double eps, double weight_decay) {
    AdamScalars k;
    k.decay = (float)(1.0 - lr * weight_decay);
    k.w1 = (float)(1.0 - beta1);
    k.b2 = (float)beta2;
    k.omb2 = (float)(1.0 - beta2);
    k.eps = (float)eps;
    return k;
}

int cvg_adamw(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
              double weight_decay, int32_t step, cvg_stream stream) {
    CVG_CHECK(p && g && m && v && n > 0 && step >= 1, "adamw: bad args");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    adamw_kernel<<<grid1(ceil_div(n, 4)), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n,
                                                                       adam_scalars(lr, beta1, beta2, eps, weight_decay),
                                                                       (float)(lr / bc1), (float)sqrt(bc2));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, int64_t* step_counter, float* bc, cvg_stream stream) {
    CVG_CHECK(p && g && m && v && n > 0 && step_counter && bc, "adamw_dev: bad args");
    hipStream_t s = (hipStream_t)stream;
    adamw_tick<<<1, 64, 0, s>>>((long long*)step_counter, bc, lr, beta1, beta2);
    CVG_LAUNCH_CHECK();
    adamw_kernel_dev<<<grid1(ceil_div(n, 4)), 256, 0, s>>>(p, g, m, v, n, adam_scalars(lr, beta1, beta2, eps, weight_decay), bc);
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_dropout_mask_dev(uint8_t* mask, int64_t n, int64_t* seed_counter, cvg_stream stream) {
    CVG_CHECK(mask && n > 0 && seed_counter, "dropout_mask_dev: bad args");
    hipStream_t s = (hipStream_t)stream;
    counter_tick<<<1, 64, 0, s>>>((long long*)seed_counter);
    CVG_LAUNCH_CHECK();
    dropout_mask_dev<<<grid1(n), 256, 0, s>>>(mask, n, (const long long*)seed_counter);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_cast(const void* in, int32_t in_dtype, void* out, int32_t out_dtype, int64_t n, cvg_stream stream) {
    CVG_CHECK(in && out && n > 0, "cast: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int g = grid1(n);
    if (in_dtype == CVG_F32 && out_dtype == CVG_BF16) cast_kernel<float, bf16><<<g, 256, 0, s>>>((const float*)in, (bf16*)out, n);
    else if (in_dtype == CVG_BF16 && out_dtype == CVG_F32) cast_kernel<bf16, float><<<g, 256, 0, s>>>((const bf16*)in, (float*)out, n);
    else if (in_dtype == CVG_F32 && out_dtype == CVG_F32) cast_kernel<float, float><<<g, 256, 0, s>>>((const float*)in, (float*)out, n);
    else if (in_dtype == CVG_BF16 && out_dtype == CVG_BF16) cast_kernel<bf16, bf16><<<g, 256, 0, s>>>((const bf16*)in, (bf16*)out, n);
    else CVG_FAIL("cast: bad dtypes");
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_nhwc_to_nchw(const void* in, int32_t dtype, float* out, int32_t B, int32_t HW, int32_t C, int32_t ld,
                     cvg_stream stream) {
    CVG_CHECK(in && out && ld >= C, "nhwc_to_nchw: bad args");
    DISPATCH(dtype, nhwc_to_nchw<T><<<grid1((int64_t)B * C * HW), 256, 0, (hipStream_t)stream>>>((const T*)in, out, B, HW, C, ld));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_nchw_to_nhwc(const float* in, void* out, int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t ld,
                     cvg_stream stream) {
    CVG_CHECK(in && out && ld >= C, "nchw_to_nhwc: bad args");
    DISPATCH(dtype, nchw_to_nhwc<T><<<grid1((int64_t)B * HW * ld), 256, 0, (hipStream_t)stream>>>(in, (T*)out, B, HW, C, ld));
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_gan_d_head(const void* h, int32_t dtype, int32_t ld, const float* w, const float* b, int32_t M, int32_t F,
                   int32_t n_real, float grad_scale, float* dw, float* db, float* loss, void* dh, float* ws,
                   cvg_stream stream) {
    CVG_CHECK(h && w && b && dw && db && loss && dh && ws && M > 0 && F > 0 && F <= 1022 && ld >= F && n_real >= 0 &&
              n_real <= M, "gan_d_head: bad args");
    hipStream_t s = (hipStream_t)stream;
    float* dlogit = ws;
    float* loss_row = ws + M;
    DISPATCH(dtype, d_head_rows<T><<<grid1(M, 4), 256, 0, s>>>((const T*)h, ld, w, b, M, F, n_real, grad_scale, dlogit,
                                                              loss_row, (T*)dh));
    CVG_LAUNCH_CHECK();
    DISPATCH(dtype, d_head_reduce<T><<<1, 1024, 0, s>>>((const T*)h, ld, dlogit, loss_row, M, F, dw, db, loss));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_gan_c_head(const void* h, int32_t dtype, int32_t ld, const float* W, const float* b, const int32_t* labels,
                   int32_t M, int32_t F, int32_t y, int32_t n_real, float grad_scale, float* dW, float* db, float* loss,
                   void* dh, float* ws, cvg_stream stream) {
    CVG_CHECK(h && W && b && labels && dW && db && loss && dh && ws && M > 0 && F > 0 && F <= 1022 && y > 0 && y <= 1024 &&
              ld >= F && n_real > 0 && n_real <= M, "gan_c_head: bad args");
    hipStream_t s = (hipStream_t)stream;
    float* dlogits = ws;
    float* loss_row = ws + (size_t)M * y;
    DISPATCH(dtype, c_head_rows<T><<<M, 256, 0, s>>>((const T*)h, ld, W, b, labels, M, F, y, n_real, grad_scale, dlogits,
                                                    loss_row, (T*)dh));
    CVG_LAUNCH_CHECK();
    DISPATCH(dtype, c_head_reduce<T><<<y, 1024, 0, s>>>((const T*)h, ld, dlogits, loss_row, F, y, n_real, dW, db, loss));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_gan_feature_match(const void* h, int32_t dtype, int32_t ld, int32_t M, int32_t F, int32_t r0, int32_t r1,
                          int32_t f0, int32_t f1, float grad_scale, float* loss, void* dh, cvg_stream stream) {
    CVG_CHECK(h && loss && dh && M > 0 && F > 0 && ld >= F && ld <= 1024 && 0 <= r0 && r0 < r1 && r1 <= M && 0 <= f0 &&
              f0 < f1 && f1 <= M, "gan_feature_match: bad args");
    DISPATCH(dtype, feature_match<T><<<1, 1024, 0, (hipStream_t)stream>>>((const T*)h, ld, M, F, r0, r1, f0, f1,
                                                                          grad_scale, loss, (T*)dh));
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_gan_class_feature_match(const void* h, int32_t dtype, int32_t ld, int32_t F, const int32_t* labels_real,
                                const int32_t* labels_fake, int32_t r0, int32_t r1, int32_t f0, int32_t f1, int32_t n_class,
                                float* centres, int32_t* seen, float momentum, float grad_scale, float* loss, void* dh,
                                float* ws, cvg_stream stream) {
    CVG_CHECK(h && labels_real && labels_fake && centres && seen && loss && dh && ws && F > 0 && ld >= F && n_class > 0 &&
              0 <= r0 && r0 < r1 && 0 <= f0 && f0 < f1, "gan_class_feature_match: bad args");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype, class_feature_match<T><<<n_class, 256, 0, s>>>((const T*)h, ld, F, labels_real, labels_fake, r0, r1, f0, f1,
                                                                   centres, seen, momentum, grad_scale, ws, (T*)dh));
    CVG_LAUNCH_CHECK();
    mse_finalize<<<1, 256, 0, s>>>(ws, n_class, loss, 1.f);
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_gan_pair_feature_match(const void* h, int32_t dtype, int32_t ld, int32_t F, int32_t r0, int32_t f0, int32_t n,
                               float grad_scale, float* loss, void* dh, float* ws, cvg_stream stream) {
    CVG_CHECK(h && loss && dh && ws && F > 0 && ld >= F && n > 0 && r0 >= 0 && f0 >= 0 && (r0 + n <= f0 || f0 + n <= r0),
              "gan_pair_feature_match: bad args");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH(dtype, pair_feature_match<T><<<n, 256, 0, s>>>((const T*)h, ld, F, r0, f0, n, grad_scale, ws, (T*)dh));
    CVG_LAUNCH_CHECK();
    mse_finalize<<<1, 256, 0, s>>>(ws, n, loss, 1.f / (float)n);
    CVG_LAUNCH_CHECK();
    return 0;
}
int cvg_axpy_f32(float* y, const float* x, float alpha, int64_t n, cvg_stream stream) {
    CVG_CHECK(y && x && n > 0, "axpy: bad args");
    axpy_f32<<<grid1(n), 256, 0, (hipStream_t)stream>>>(y, x, alpha, n);
    CVG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
