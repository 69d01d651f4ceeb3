// HBM-bound kernels of the StackGAN-v2 step on gfx950: BatchNorm statistics/apply fused with
// GLU / LeakyReLU / residual add, their backward passes (two per-channel reductions + apply),
// layout conversion, CA_NET reparameterisation + KL, logit heads + BCE, class-aware loss,
// fused Adam / EMA.  All tensors NHWC fp32; every thread moves float4 (16 B/lane).
#include "s2i_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }
// the gate of the GLU passes over bf16 tensors: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x2,
// v_rcp, four v_fma, v_div_fmas, v_div_fixup): those passes are VALU-bound (tools/elementwise_bench.py: GLU backward
// reduce 1.75 -> 2.64 TB/s), and the gate's relative error stays ~1e-7, far below the bf16 rounding of its operands.
// fp32 tensors, the logit heads and the LSTM keep the exact form (the fp32 parity tests sit on LeakyReLU decisions that a
// 1e-7 perturbation re-rolls).
struct bf16_t;
template <typename T> __device__ __forceinline__ float sigmoid_gate_(float v) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
  else return 1.f / (1.f + __expf(-v));
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// activation tensors are fp32 or bf16 (bf16 activation mode, BASELINE config 4): T = float | bf16_t; the arithmetic of
// every kernel below stays fp32, only the HBM representation changes
struct bf16_t { unsigned short v; };
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
  const u32x2 h = *reinterpret_cast<const u32x2*>(p);
  return f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
               __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
}
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
  const f32x2_ a = {v[0], v[1]}, b = {v[2], v[3]};
  *reinterpret_cast<u32x2*>(p) = u32x2{__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_)),
                                       __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_))};
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __builtin_bit_cast(float, (unsigned)p->v << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { p->v = __builtin_bit_cast(unsigned short, (__bf16)v); }

// thread layout for per-channel reductions over the rows of an [M][C] tensor:
// `cpb` threads across channel quads, 256/cpb row lanes.
struct RedGeom {
  int Q, cpb, rpb, gy;
};
static RedGeom red_geom(int C) {
  RedGeom g;
  g.Q = C / 4;
  int cpb = 1;
  while (cpb < g.Q && cpb < 256) cpb <<= 1;
  g.cpb = cpb;
  g.rpb = 256 / cpb;
  g.gy = (g.Q + cpb - 1) / cpb;
  return g;
}

// ---- per-quad value functors ------------------------------------------------------------------
// dz for BN-channel quad `quad` at `row`, un-doing the activation that followed BatchNorm
template <typename T>
__device__ __forceinline__ f32x4 act_dz(const T* __restrict__ y, const T* __restrict__ dout, int lddout,
                                        long long row, int C, int quad, const float* __restrict__ coef,
                                        int act, f32x4 yv) {
  const float* scale = coef + 2 * C;
  const float* shift = coef + 3 * C;
  f32x4 dz;
  if (act == S2I_ACT_GLU) {
    const int hq = C / 8;  // quads per half
    const bool first = quad < hq;
    const int pq = first ? quad + hq : quad - hq;
    const f32x4 yp = ld4(y + row * C + pq * 4);
    const f32x4 d = ld4(dout + row * lddout + (first ? quad : pq) * 4);
    const f32x4 sa = ld4(scale + (first ? quad : pq) * 4), ta = ld4(shift + (first ? quad : pq) * 4);
    const f32x4 sg = ld4(scale + (first ? pq : quad) * 4), tg = ld4(shift + (first ? pq : quad) * 4);
    const f32x4 ya = first ? yv : yp, yg = first ? yp : yv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float za = sa[j] * ya[j] + ta[j];
      const float sgm = sigmoid_gate_<T>(sg[j] * yg[j] + tg[j]);
      dz[j] = first ? d[j] * sgm : d[j] * za * sgm * (1.f - sgm);
    }
  } else {
    const f32x4 d = ld4(dout + row * lddout + quad * 4);
    if (act == S2I_ACT_LRELU) {
      const f32x4 sc = ld4(scale + quad * 4), sh = ld4(shift + quad * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) dz[j] = (sc[j] * yv[j] + sh[j]) > 0.f ? d[j] : 0.2f * d[j];
    } else {
      dz = d;
    }
  }
  return dz;
}

// ACT >= 0: the activation as a compile-time constant (the runtime form compiles every activation's path into one kernel:
// 127 registers = 4 waves per SIMD; specialised, the LeakyReLU form needs far fewer), RPT rows per trip
template <int MODE, typename T, int ACT = -1, int RPT = 4>  // 0: (y, y^2)   1: (dz, dz*xhat)
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ y, int ldy,
                                                        const T* __restrict__ dout, int lddout,
                                                        long long M, int C, const float* __restrict__ coef,
                                                        int act_rt, float* __restrict__ part, int nparts, int cpb,
                                                        int ppg, long long Rg) {
  const int act = ACT >= 0 ? ACT : act_rt;
  // rows are split into groups of Rg rows (independent BatchNorm batches); part p covers a row chunk of
  // group p / ppg and uses that group's coefficients
  __shared__ f32x4 sh[2][256];
  const int tid = threadIdx.x;
  const int rpb = 256 / cpb;
  const int ql = tid % cpb, rl = tid / cpb;
  const int quad = blockIdx.y * cpb + ql;
  const int Q = C / 4;
  const int grp = blockIdx.x / ppg, pp = blockIdx.x - grp * ppg;
  const long long chunk = (Rg + ppg - 1) / ppg;
  const long long r0 = grp * Rg + pp * chunk;
  const long long gend = (grp + 1) * Rg < M ? (grp + 1) * Rg : M;
  const long long r1 = r0 + chunk < gend ? r0 + chunk : gend;
  coef += (size_t)grp * 4 * C;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (quad < Q) {
    f32x4 mean = {0.f, 0.f, 0.f, 0.f}, invstd = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) { mean = ld4(coef + quad * 4); invstd = ld4(coef + C + quad * 4); }
    // four rows per trip: their loads are issued together (one row per trip left ~2 loads per lane in flight: 2.9 TB/s)
    long long row = r0 + rl;
    for (; row + (RPT - 1) * rpb < r1; row += RPT * rpb) {
      f32x4 yv[RPT], dz[RPT];
#pragma unroll
      for (int u = 0; u < RPT; ++u) yv[u] = ld4(y + (row + u * rpb) * ldy + quad * 4);
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) dz[u] = act_dz(y, dout, lddout, row + u * rpb, C, quad, coef, act, yv[u]);
      }
#pragma unroll
      for (int u = 0; u < RPT; ++u) {
        if (MODE == 0) {
          s0 += yv[u];
          s1 += yv[u] * yv[u];
        } else {
          s0 += dz[u];
          s1 += dz[u] * ((yv[u] - mean) * invstd);
        }
      }
    }
    for (; row < r1; row += rpb) {
      const f32x4 yv = ld4(y + row * ldy + quad * 4);
      if (MODE == 0) {
        s0 += yv;
        s1 += yv * yv;
      } else {
        const f32x4 dz = act_dz(y, dout, lddout, row, C, quad, coef, act, yv);
        s0 += dz;
        s1 += dz * ((yv - mean) * invstd);
      }
    }
  }
  sh[0][tid] = s0;
  sh[1][tid] = s1;
  __syncthreads();
  if (rl == 0 && quad < Q) {
    for (int r = 1; r < rpb; ++r) {
      s0 += sh[0][r * cpb + ql];
      s1 += sh[1][r * cpb + ql];
    }
    st4(part + ((size_t)0 * nparts + blockIdx.x) * C + quad * 4, s0);
    st4(part + ((size_t)1 * nparts + blockIdx.x) * C + quad * 4, s1);
  }
}

// Reduce [2][G*ppg][C] partials in double.  Threads = qpb channel quads x `lanes` row lanes (tid = pl * qpb + ql); the G
// groups (independent BatchNorm batches sharing one set of parameters) own contiguous ranges of `lpg` row lanes and are
// reduced AT THE SAME TIME: per-lane sums, a shuffle reduction inside each wave over the lanes of equal quad (no barrier),
// one LDS exchange between waves, then one thread per quad walks the groups in order -- so the running statistics see G
// successive momentum updates exactly as G separate forwards would give.  (The earlier form ran a 10-level LDS tree with
// a barrier per level, once per group: 17 us for what is a few hundred KB.)
template <int MODE>  // 0: BN forward statistics   1: BN backward sums
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ part, int ppg, int G, int C,
                                                           double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ rmean,
                                                           float* __restrict__ rvar, float momentum, float eps,
                                                           float* __restrict__ out, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int accumulate, int qpb, int lpg,
                                                           long long* __restrict__ nbt) {
  extern __shared__ double shd[];  // [waves][qpb][8] partial sums, then [G][qpb][8] group sums
  if (MODE == 0 && nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += G;  // num_batches_tracked
  const int tid = threadIdx.x;
  const int NT = blockDim.x;
  const int ql = tid % qpb, pl = tid / qpb;
  const int quad = blockIdx.x * qpb + ql;
  const int Q = C / 4;
  const int nparts = ppg * G;
  const int grp = pl / lpg, pin = pl - grp * lpg;   // this lane's group and its lane index inside the group
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (quad < Q && grp < G) {
    for (int pi = grp * ppg + pin; pi < (grp + 1) * ppg; pi += lpg) {
      const f32x4 v0 = ld4(part + ((size_t)0 * nparts + pi) * C + quad * 4);
      const f32x4 v1 = ld4(part + ((size_t)1 * nparts + pi) * C + quad * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] += v0[j]; a[4 + j] += v1[j]; }
    }
  }
  // lanes of one wave that share the quad AND the group: xor offsets qpb .. 32 stay inside a group when lpg * qpb >= 64
  // (a group then covers whole waves); smaller blocks take the LDS path only
  const int wave = tid >> 6, lane = tid & 63, nwaves = (NT + 63) >> 6;
  const bool whole_waves = (lpg * qpb) % 64 == 0;
  if (whole_waves) {
    for (int off = 32; off >= qpb; off >>= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += __shfl_xor(a[j], off);
    }
    if (lane < qpb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) shd[((size_t)wave * qpb + lane) * 8 + j] = a[j];
    }
  } else {
    // few threads per group inside one wave: every lane publishes, the group leader sums
#pragma unroll
    for (int j = 0; j < 8; ++j) shd[(size_t)tid * 8 + j] = a[j];
  }
  __syncthreads();
  // group sums -> shd2[g][ql][8] (kept in registers of the group's first lane, then exchanged)
  double gs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (pin == 0 && grp < G && quad < Q) {
    if (whole_waves) {
      const int w0 = (grp * lpg * qpb) >> 6, w1 = (((grp + 1) * lpg * qpb) + 63) >> 6;
      for (int w = w0; w < w1 && w < nwaves; ++w)
#pragma unroll
        for (int j = 0; j < 8; ++j) gs[j] += shd[((size_t)w * qpb + ql) * 8 + j];
    } else {
      for (int k = 0; k < lpg; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) gs[j] += shd[((size_t)((grp * lpg + k) * qpb + ql)) * 8 + j];
    }
  }
  __syncthreads();
  if (pin == 0 && grp < G && quad < Q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) shd[((size_t)grp * qpb + ql) * 8 + j] = gs[j];
  }
  __syncthreads();
  if (pl != 0 || quad >= Q) return;
  double g0[4] = {0, 0, 0, 0}, g1[4] = {0, 0, 0, 0};  // sums over groups (backward: dbeta, dgamma)
  float rm[4] = {0.f, 0.f, 0.f, 0.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0 && rmean) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { rm[j] = rmean[quad * 4 + j]; rv[j] = rvar[quad * 4 + j]; }
  }
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = quad * 4 + j;
      const double a0 = shd[((size_t)g * qpb + ql) * 8 + j], a1 = shd[((size_t)g * qpb + ql) * 8 + 4 + j];
      if (MODE == 0) {
        float* o = out + (size_t)g * 4 * C;
        const double mean = a0 / count;
        double var = a1 / count - mean * mean;
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        o[c] = (float)mean;
        o[C + c] = invstd;
        o[2 * C + c] = sc;
        o[3 * C + c] = beta[c] - (float)mean * sc;
        const double unb = count > 1 ? var * count / (count - 1) : var;
        rm[j] = (1.f - momentum) * rm[j] + momentum * (float)mean;
        rv[j] = (1.f - momentum) * rv[j] + momentum * (float)unb;
      } else {
        float* o = out + (size_t)g * 2 * C;
        o[c] = (float)(a0 / count);
        o[C + c] = (float)(a1 / count);
        g0[j] += a0;
        g1[j] += a1;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = quad * 4 + j;
    if (MODE == 0) {
      if (rmean) { rmean[c] = rm[j]; rvar[c] = rv[j]; }
    } else {
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)g0[j] : (float)g0[j];
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)g1[j] : (float)g1[j];
    }
  }
}

// Same result for very short partial lists (<= 8 rows per group): one thread per channel, no LDS, no barriers.
template <int MODE>
__global__ __launch_bounds__(256) void bn_finalize_small_kernel(const float* __restrict__ part, int ppg, int G, int C,
                                                                double count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, float momentum, float eps,
                                                                float* __restrict__ out, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int accumulate,
                                                                long long* __restrict__ nbt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (MODE == 0 && nbt && c == 0) nbt[0] += G;
  if (c >= C) return;
  const int nparts = ppg * G;
  double g0 = 0, g1 = 0;
  float rm = 0.f, rv = 0.f;
  if (MODE == 0 && rmean) { rm = rmean[c]; rv = rvar[c]; }
  for (int grp = 0; grp < G; ++grp) {
    double a0 = 0, a1 = 0;
    for (int pi = grp * ppg; pi < (grp + 1) * ppg; ++pi) {
      a0 += part[((size_t)0 * nparts + pi) * C + c];
      a1 += part[((size_t)1 * nparts + pi) * C + c];
    }
    if (MODE == 0) {
      float* o = out + (size_t)grp * 4 * C;
      const double mean = a0 / count;
      double var = a1 / count - mean * mean;
      if (var < 0) var = 0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = gamma[c] * invstd;
      o[c] = (float)mean;
      o[C + c] = invstd;
      o[2 * C + c] = sc;
      o[3 * C + c] = beta[c] - (float)mean * sc;
      const double unb = count > 1 ? var * count / (count - 1) : var;
      rm = (1.f - momentum) * rm + momentum * (float)mean;
      rv = (1.f - momentum) * rv + momentum * (float)unb;
    } else {
      float* o = out + (size_t)grp * 2 * C;
      o[c] = (float)(a0 / count);
      o[C + c] = (float)(a1 / count);
      g0 += a0;
      g1 += a1;
    }
  }
  if (MODE == 0) {
    if (rmean) { rmean[c] = rm; rvar[c] = rv; }
  } else {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)g0 : (float)g0;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)g1 : (float)g1;
  }
}

__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rvar[c] + eps);
  const float sc = gamma[c] * invstd;
  out[c] = rmean[c];
  out[C + c] = invstd;
  out[2 * C + c] = sc;
  out[3 * C + c] = beta[c] - rmean[c] * sc;
}

template <typename T, int ACT = -1>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, long long M, int C,
                                                         const float* __restrict__ coef0, int act_rt,
                                                         const T* __restrict__ residual,
                                                         T* __restrict__ out, int G, unsigned Rg) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int Cout = act == S2I_ACT_GLU ? C / 2 : C;
  const int Qo = Cout / 4;
  const long long total = M * Qo;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / Qo;
    const int q = (int)(e - row * Qo);
    const float* coef = G > 1 ? coef0 + (size_t)((unsigned)row / Rg) * 4 * C : coef0;
    const float* scale = coef + 2 * C;
    const float* shift = coef + 3 * C;
    f32x4 o;
    if (act == S2I_ACT_GLU) {
      const f32x4 ya = ld4(y + row * C + q * 4), yg = ld4(y + row * C + Cout + q * 4);
      const f32x4 sa = ld4(scale + q * 4), ta = ld4(shift + q * 4);
      const f32x4 sg = ld4(scale + Cout + q * 4), tg = ld4(shift + Cout + q * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (sa[j] * ya[j] + ta[j]) * sigmoid_gate_<T>(sg[j] * yg[j] + tg[j]);
    } else {
      const f32x4 yv = ld4(y + row * C + q * 4);
      const f32x4 sc = ld4(scale + q * 4), sh = ld4(shift + q * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = sc[j] * yv[j] + sh[j];
        if (act == S2I_ACT_LRELU) z = z > 0.f ? z : 0.2f * z;
        o[j] = z;
      }
      if (residual) o += ld4(residual + row * C + q * 4);
    }
    st4(out + row * Cout + q * 4, o);
  }
}

template <typename T, int ACT = -1>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ y,
                                                               const T* __restrict__ dout, int lddout,
                                                               long long M, int C, const float* __restrict__ coef0,
                                                               const float* __restrict__ red20, int act_rt,
                                                               T* __restrict__ dy, int G, unsigned Rg) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int Q = C / 4;
  const long long total = M * Q;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / Q;
    const int q = (int)(e - row * Q);
    const unsigned grp = G > 1 ? (unsigned)row / Rg : 0u;
    const float* coef = coef0 + (size_t)grp * 4 * C;
    const float* red2 = red20 + (size_t)grp * 2 * C;
    const f32x4 yv = ld4(y + row * C + q * 4);
    const f32x4 dz = act_dz(y, dout, lddout, row, C, q, coef, act, yv);
    const f32x4 mean = ld4(coef + q * 4), invstd = ld4(coef + C + q * 4), sc = ld4(coef + 2 * C + q * 4);
    const f32x4 m0 = ld4(red2 + q * 4), m1 = ld4(red2 + C + q * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (yv[j] - mean[j]) * invstd[j];
      o[j] = sc[j] * (dz[j] - m0[j] - xh * m1[j]);
    }
    st4(dy + row * C + q * 4, o);
  }
}

// The backward apply as a WALKER (round 2): a thread keeps one channel quad and walks the rows of its block's chunk, as
// colreduce_kernel does, with EVERY coefficient in registers (loaded by hand before the row loop: the stores to dy inside
// the loop keep the compiler from hoisting them).  The grid-stride form above re-loads seven to nine 16-byte coefficient
// vectors per quad -- L1 hits, but 168 bytes through the CU's 64 B/clk vector-memory path for 24 bytes of data: on the
// 32-channel GLU tensors of the generator that path, not HBM, set the 2.3 TB/s.
template <typename T, int ACT, int RPT = 4>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_walk_kernel(const T* __restrict__ y, const T* __restrict__ dout,
                                                                    int lddout, long long M, int C,
                                                                    const float* __restrict__ coef0,
                                                                    const float* __restrict__ red20, T* __restrict__ dy,
                                                                    int cpb, int ppg, long long Rg) {
  const int tid = threadIdx.x;
  const int rpb = 256 / cpb;
  const int ql = tid % cpb, rl = tid / cpb;
  const int quad = blockIdx.y * cpb + ql;
  if (quad >= C / 4) return;
  const int grp = blockIdx.x / ppg, pp = blockIdx.x - grp * ppg;
  const long long chunk = (Rg + ppg - 1) / ppg;
  const long long r0 = grp * Rg + pp * chunk;
  const long long gend = (grp + 1) * Rg < M ? (grp + 1) * Rg : M;
  const long long r1 = r0 + chunk < gend ? r0 + chunk : gend;
  const float* coef = coef0 + (size_t)grp * 4 * C;
  const float* red2 = red20 + (size_t)grp * 2 * C;
  const float* scale = coef + 2 * C;
  const float* shift = coef + 3 * C;
  const f32x4 mean = ld4(coef + quad * 4), invstd = ld4(coef + C + quad * 4), sc = ld4(scale + quad * 4);
  const f32x4 sh = ld4(shift + quad * 4);
  const f32x4 m0 = ld4(red2 + quad * 4), m1 = ld4(red2 + C + quad * 4);
  // GLU: this quad is in the value half (first) or the gate half; pq is its partner quad in the other half
  const int hq = C / 8;
  const bool first = quad < hq;
  const int pq = ACT == S2I_ACT_GLU ? (first ? quad + hq : quad - hq) : quad;
  const int dq = ACT == S2I_ACT_GLU ? (first ? quad : pq) : quad;          // quad of dout
  f32x4 sp = sc, tp = sh;                                                   // partner's scale / shift
  if (ACT == S2I_ACT_GLU) { sp = ld4(scale + pq * 4); tp = ld4(shift + pq * 4); }
  auto finish = [&](long long row, const f32x4& yv, const f32x4& yp, const f32x4& d) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float dz;
      if (ACT == S2I_ACT_GLU) {
        // value half: dz = d * sigmoid(gate);  gate half: dz = d * value * sigmoid(gate) * (1 - sigmoid(gate))
        const float za = first ? sc[j] * yv[j] + sh[j] : sp[j] * yp[j] + tp[j];
        const float zg = first ? sp[j] * yp[j] + tp[j] : sc[j] * yv[j] + sh[j];
        const float sgm = sigmoid_gate_<T>(zg);
        dz = first ? d[j] * sgm : d[j] * za * sgm * (1.f - sgm);
      } else if (ACT == S2I_ACT_LRELU) {
        dz = (sc[j] * yv[j] + sh[j]) > 0.f ? d[j] : 0.2f * d[j];
      } else {
        dz = d[j];
      }
      const float xh = (yv[j] - mean[j]) * invstd[j];
      o[j] = sc[j] * (dz - m0[j] - xh * m1[j]);
    }
    st4(dy + row * C + quad * 4, o);
  };
  long long row = r0 + rl;
  for (; row + (RPT - 1) * rpb < r1; row += RPT * rpb) {
    f32x4 yv[RPT], yp[RPT], d[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      yv[u] = ld4(y + (row + u * rpb) * C + quad * 4);
      d[u] = ld4(dout + (row + u * rpb) * lddout + dq * 4);
      yp[u] = ACT == S2I_ACT_GLU ? ld4(y + (row + u * rpb) * C + pq * 4) : yv[u];
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) finish(row + u * rpb, yv[u], yp[u], d[u]);
  }
  for (; row < r1; row += rpb) {
    const f32x4 yv = ld4(y + row * C + quad * 4);
    const f32x4 d = ld4(dout + row * lddout + dq * 4);
    const f32x4 yp = ACT == S2I_ACT_GLU ? ld4(y + row * C + pq * 4) : yv;
    finish(row, yv, yp, d);
  }
}

template <typename T, int ACT = -1>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ out, const T* __restrict__ dout,
                                                      int lddout, long long M, int C, int act_rt,
                                                      T* __restrict__ dy) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int Q = C / 4;
  const long long total = M * Q;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / Q;
    const int q = (int)(e - row * Q);
    const f32x4 ov = ld4(out + row * C + q * 4);
    const f32x4 d = ld4(dout + row * lddout + q * 4);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (act == S2I_ACT_LRELU) o[j] = ov[j] > 0.f ? d[j] : 0.2f * d[j];
      else if (act == S2I_ACT_TANH) o[j] = d[j] * (1.f - ov[j] * ov[j]);
      else o[j] = d[j];
    }
    st4(dy + row * C + q * 4, o);
  }
}

__global__ void glu_fwd_kernel(const float* __restrict__ x, long long M, int C, float* __restrict__ out) {
  const int H = C / 2;
  const long long total = M * H;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / H;
    const int c = (int)(e - row * H);
    out[e] = x[row * C + c] * sigmoidf_(x[row * C + H + c]);
  }
}

__global__ void glu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout, long long M, int C,
                               float* __restrict__ dx) {
  const int H = C / 2;
  const long long total = M * H;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / H;
    const int c = (int)(e - row * H);
    const float a = x[row * C + c];
    const float sg = sigmoidf_(x[row * C + H + c]);
    const float d = dout[e];
    dx[row * C + c] = d * sg;
    dx[row * C + H + c] = d * a * sg * (1.f - sg);
  }
}

// ---- layout -----------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int HW,
                                    int Cp) {
  const long long total = (long long)B * HW * Cp;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % Cp);
    const long long bp = e / Cp;
    const int pix = (int)(bp % HW);
    const int b = (int)(bp / HW);
    st1(dst + e, c < C ? src[((long long)b * C + c) * HW + pix] : 0.f);
  }
}
// image fast path: C = 3 -> Cp = 4, one pixel per thread
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int HW) {
  const long long total = (long long)B * HW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int pix = (int)(e % HW);
    const long long b = e / HW;
    const float* s = src + b * 3 * HW + pix;
    f32x4 v = {s[0], s[HW], s[2 * (long long)HW], 0.f};
    st4(dst + e * 4, v);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int lds, float* __restrict__ dst, int B, int C,
                                    int HW) {
  const long long total = (long long)B * C * HW;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int pix = (int)(e % HW);
    const long long bc = e / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    dst[e] = ld1(src + (b * HW + pix) * lds + c);
  }
}

__global__ void image_to_u8_kernel(const float* __restrict__ src, int lds, unsigned char* __restrict__ dst,
                                   long long npix) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < npix;
       e += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = ((src[e * lds + c] + 1.f) / 2.f) * 255.f;
      v = fminf(fmaxf(v, 0.f), 255.f);
      dst[e * 3 + c] = (unsigned char)v;  // .byte(): truncation
    }
  }
}

// HWC uint8 -> normalised NCHW float: consecutive threads take consecutive pixels, so the 3-byte reads and the three
// plane writes are all coalesced
__global__ void u8_to_image_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int HW,
                                   long long npix) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < npix;
       e += (long long)gridDim.x * blockDim.x) {
    const long long b = e / HW;
    const long long r = e - b * HW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = (float)src[e * 3 + c] / 255.f;          // ToTensor
      dst[(b * 3 + c) * HW + r] = (t - 0.5f) / 0.5f;          // Normalize
    }
  }
}

// per-image column sums, two stages: [B][S][C] partials then the S-sum
template <typename T>
__global__ __launch_bounds__(256) void spatial_sum_stage1(const T* __restrict__ src, int ld, int HW, int C,
                                                          int S, float* __restrict__ tmp, int cpb) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x;
  const int rpb = 256 / cpb;
  const int ql = tid % cpb, rl = tid / cpb;
  const int quad = blockIdx.z * cpb + ql;
  const int Q = C / 4;
  const int b = blockIdx.x, sidx = blockIdx.y;
  const int chunk = (HW + S - 1) / S;
  const int r0 = sidx * chunk, r1 = min(HW, r0 + chunk);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (quad < Q)
    for (int r = r0 + rl; r < r1; r += rpb) acc += ld4(src + ((long long)b * HW + r) * ld + quad * 4);
  sh[tid] = acc;
  __syncthreads();
  if (rl == 0 && quad < Q) {
    for (int r = 1; r < rpb; ++r) acc += sh[r * cpb + ql];
    st4(tmp + ((size_t)b * S + sidx) * C + quad * 4, acc);
  }
}
__global__ void spatial_sum_stage2(const float* __restrict__ tmp, int B, int S, int C, float* __restrict__ dst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int b = e / C, c = e - b * C;
  float v = 0.f;
  for (int s = 0; s < S; ++s) v += tmp[((size_t)b * S + s) * C + c];
  dst[e] = v;
}

// ---- spatially constant channels of a 3x3 conv --------------------------------------------------
// valid taps of border class cls = 3*ry + rx (ry: 0 top, 1 middle, 2 bottom): ky in [ry==0, 2-(ry==2)]
// one block per (image, tap): T[b][t][n] = sum_cc c[b][cc] * P[t][cc][n]; c in LDS, n across threads
__global__ __launch_bounds__(256) void cvec_tap_table_kernel(const float* __restrict__ cvec,
                                                             const float* __restrict__ packed, int Cc, int Ip, int Op,
                                                             int N, float* __restrict__ taps) {
  extern __shared__ float cs[];
  const int b = blockIdx.x / 9, t = blockIdx.x - b * 9;
  for (int i = threadIdx.x; i < Cc; i += 256) cs[i] = cvec[b * Cc + i];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {
    const float* wp = packed + ((size_t)t * Ip) * Op + n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int cc = 0;
    for (; cc + 3 < Cc; cc += 4) {
      a0 += cs[cc] * wp[(size_t)cc * Op];
      a1 += cs[cc + 1] * wp[(size_t)(cc + 1) * Op];
      a2 += cs[cc + 2] * wp[(size_t)(cc + 2) * Op];
      a3 += cs[cc + 3] * wp[(size_t)(cc + 3) * Op];
    }
    for (; cc < Cc; ++cc) a0 += cs[cc] * wp[(size_t)cc * Op];
    taps[((size_t)b * 9 + t) * N + n] = (a0 + a1) + (a2 + a3);
  }
}
// valid taps of border class cls = 3*ry + rx (ry: 0 top, 1 middle, 2 bottom): ky in [ry==0, 2-(ry==2)]
__global__ void cvec_bias_table_kernel(const float* __restrict__ taps, int B, int N, float* __restrict__ table) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * 9 * N) return;
  const int n = e % N;
  const int cls = (e / N) % 9;
  const int b = e / (9 * N);
  const int ry = cls / 3, rx = cls - ry * 3;
  const int ky0 = ry == 0 ? 1 : 0, ky1 = ry == 2 ? 1 : 2, kx0 = rx == 0 ? 1 : 0, kx1 = rx == 2 ? 1 : 2;
  float acc = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) acc += taps[((size_t)b * 9 + ky * 3 + kx) * N + n];
  table[e] = acc;
}

// stage 1: per image and row band, 9 border sums per channel:
//   0 total, 1 row y=0, 2 row y=H-1, 3 col x=0, 4 col x=W-1, 5..8 corners (0,0) (0,W-1) (H-1,0) (H-1,W-1)
template <typename T>
__global__ __launch_bounds__(256) void border_sums_stage1(const T* __restrict__ dy, int H, int W, int C, int S,
                                                          float* __restrict__ tmp, int cpb) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x;
  const int rpb = 256 / cpb;
  const int ql = tid % cpb, rl = tid / cpb;
  const int quad = blockIdx.z * cpb + ql;
  const int Q = C / 4;
  const int b = blockIdx.x, sidx = blockIdx.y;
  const int band = (H + S - 1) / S;
  const int y0 = sidx * band, y1 = min(H, y0 + band);
  f32x4 a[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) a[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (quad < Q) {
    for (int y = y0; y < y1; ++y) {
      const bool top = y == 0, bot = y == H - 1;
      for (int x = rl; x < W; x += rpb) {
        const f32x4 v = ld4(dy + (((long long)b * H + y) * W + x) * C + quad * 4);
        const bool lft = x == 0, rgt = x == W - 1;
        a[0] += v;
        if (top) a[1] += v;
        if (bot) a[2] += v;
        if (lft) a[3] += v;
        if (rgt) a[4] += v;
        if (top && lft) a[5] += v;
        if (top && rgt) a[6] += v;
        if (bot && lft) a[7] += v;
        if (bot && rgt) a[8] += v;
      }
    }
  }
  for (int k = 0; k < 9; ++k) {
    __syncthreads();
    sh[tid] = a[k];
    __syncthreads();
    if (rl == 0 && quad < Q) {
      f32x4 v = a[k];
      for (int r = 1; r < rpb; ++r) v += sh[r * cpb + ql];
      st4(tmp + (((size_t)b * S + sidx) * 9 + k) * C + quad * 4, v);
    }
  }
}
// stage 2: sum the bands, then tapsum[b][t][c] = total - excluded row - excluded col + excluded corner
__global__ void border_sums_stage2(const float* __restrict__ tmp, int B, int S, int C, float* __restrict__ tapsum) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int b = e / C, c = e - b * C;
  float v[9];
  for (int k = 0; k < 9; ++k) {
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += tmp[(((size_t)b * S + s) * 9 + k) * C + c];
    v[k] = acc;
  }
  for (int ky = 0; ky < 3; ++ky)
    for (int kx = 0; kx < 3; ++kx) {
      // tap (ky,kx) is out of bounds on row y=0 when ky==0, on row H-1 when ky==2; same for columns
      const int er = ky == 0 ? 1 : (ky == 2 ? 2 : 0), ec = kx == 0 ? 3 : (kx == 2 ? 4 : 0);
      float r = v[0];
      if (er) r -= v[er];
      if (ec) r -= v[ec];
      if (er && ec) r += v[5 + (er == 2 ? 2 : 0) + (ec == 4 ? 1 : 0)];
      tapsum[((size_t)b * 9 + ky * 3 + kx) * C + c] = r;
    }
}
// dc[b][cc] = sum_{t,n} P[t][cc][n] * tapsum[b][t][n]: block = 64 cc x 4 lanes over the 9*N products
__global__ __launch_bounds__(256) void cvec_dc_kernel(const float* __restrict__ packed, const float* __restrict__ tapsum,
                                                      int B, int Cc, int Ip, int Op, int N, float* __restrict__ dc) {
  __shared__ float sh[256];
  const int b = blockIdx.x, cl = threadIdx.x & 63, ln = threadIdx.x >> 6;
  const int cc = blockIdx.y * 64 + cl;
  float acc = 0.f;
  if (cc < Cc) {
    for (int t = 0; t < 9; ++t) {
      const float* wp = packed + ((size_t)t * Ip + cc) * Op;
      const float* sp = tapsum + ((size_t)b * 9 + t) * N;
      for (int n = ln * 4; n + 3 < N; n += 16) {
        const f32x4 w4 = ld4(wp + n), s4 = ld4(sp + n);
        acc += w4[0] * s4[0] + w4[1] * s4[1] + w4[2] * s4[2] + w4[3] * s4[3];
      }
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (ln == 0 && cc < Cc) dc[b * Cc + cc] = (sh[cl] + sh[64 + cl]) + (sh[128 + cl] + sh[192 + cl]);
}
__global__ void cvec_dw_kernel(const float* __restrict__ cvec, const float* __restrict__ tapsum, int B, int Cc, int N,
                               int O, int I_total, float* __restrict__ dw, int accumulate) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= O * Cc * 9) return;
  const int t = e % 9;
  const int cc = (e / 9) % Cc;
  const int o = e / (9 * Cc);
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += cvec[b * Cc + cc] * tapsum[((size_t)b * 9 + t) * N + o];
  float* gp = dw + ((size_t)o * I_total + cc) * 9 + t;
  *gp = accumulate ? *gp + acc : acc;
}

// ---- CA_NET ------------------------------------------------------------------------------------
__global__ void reparam_fwd_kernel(const float* __restrict__ h, const float* __restrict__ eps, int B, int E,
                                   float* __restrict__ c) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * E) return;
  const int b = e / E, j = e - b * E;
  const float mu = h[b * 2 * E + j], lv = h[b * 2 * E + E + j];
  c[e] = eps[e] * __expf(0.5f * lv) + mu;
}
__global__ void reparam_bwd_kernel(const float* __restrict__ h, const float* __restrict__ eps,
                                   const float* __restrict__ dc, const float* __restrict__ dmu,
                                   const float* __restrict__ dlv, int B, int E, float* __restrict__ dh) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * E) return;
  const int b = e / E, j = e - b * E;
  const float lv = h[b * 2 * E + E + j];
  const float g = dc ? dc[e] : 0.f;
  dh[b * 2 * E + j] = g + (dmu ? dmu[e] : 0.f);
  dh[b * 2 * E + E + j] = g * eps[e] * 0.5f * __expf(0.5f * lv) + (dlv ? dlv[e] : 0.f);
}
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, int ldmu,
                                                     const float* __restrict__ lv, int ldlv, int B, int E,
                                                     float* __restrict__ kl) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (int e = threadIdx.x; e < B * E; e += 256) {
    const int b = e / E, j = e - b * E;
    const float m = mu[b * ldmu + j], l = lv[b * ldlv + j];
    acc += 1.f + l - m * m - __expf(l);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) kl[0] = -0.5f * sh[0] / (float)(B * E);
}
__global__ void kl_bwd_kernel(const float* __restrict__ mu, int ldmu, const float* __restrict__ lv, int ldlv, int B,
                              int E, const float* __restrict__ gout, float* __restrict__ dmu,
                              float* __restrict__ dlv) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * E) return;
  const int b = e / E, j = e - b * E;
  const float g = gout[0] * (-0.5f) / (float)(B * E);
  dmu[e] = g * (-2.f * mu[b * ldmu + j]);
  dlv[e] = g * (1.f - __expf(lv[b * ldlv + j]));
}

// ---- logit heads + BCE ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logit_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int C,
                                                        float* __restrict__ prob) {
  __shared__ float sh[256];
  const int b = blockIdx.x;
  const int n = 16 * C;
  float acc = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int pix = e / C, c = e - pix * C;
    acc += x[(size_t)b * n + e] * w[c * 16 + pix];
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) prob[b] = sigmoidf_(sh[0] + (bias ? bias[0] : 0.f));
}
__global__ __launch_bounds__(256) void logit_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ prob, const float* __restrict__ dprob,
                                                        int B, int C, float* __restrict__ dx, int acc_dx,
                                                        float* __restrict__ dw, float* __restrict__ dbias, int acc_dw) {
  // dl[b] = d loss / d logit once per block; then every thread owns one (pixel, channel) column of x and walks the
  // batch eight rows at a time so that eight independent loads are in flight (the serial walk was latency-bound: 39 us)
  __shared__ float dl_s[256];
  const int n = 16 * C;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = e < n;
  int pix = 0, c = 0;
  if (live) { pix = e / C; c = e - pix * C; }
  const float wv = live ? w[c * 16 + pix] : 0.f;
  float gw = 0.f, gb = 0.f;
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int nb = min(256, B - b0);
    __syncthreads();
    if ((int)threadIdx.x < nb) {
      const float pr = prob[b0 + threadIdx.x];
      dl_s[threadIdx.x] = dprob[b0 + threadIdx.x] * pr * (1.f - pr);
    }
    __syncthreads();
    if (e == 0 && dbias)
      for (int b = 0; b < nb; ++b) gb += dl_s[b];
    if (!live) continue;
    for (int b1 = 0; b1 < nb; b1 += 8) {
      float xv[8], dv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = b1 + j < nb;
        const size_t off = (size_t)(b0 + b1 + j) * n + e;
        xv[j] = ok ? x[off] : 0.f;
        dv[j] = (ok && dx && acc_dx) ? dx[off] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (b1 + j >= nb) break;
        const float dl = dl_s[b1 + j];
        gw += dl * xv[j];
        if (dx) dx[(size_t)(b0 + b1 + j) * n + e] = dv[j] + dl * wv;
      }
    }
  }
  if (live && dw) dw[c * 16 + pix] = acc_dw ? dw[c * 16 + pix] + gw : gw;
  if (e == 0 && dbias) dbias[0] = acc_dw ? dbias[0] + gb : gb;
}
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ prob, float target, int B,
                                                      float weight, float* __restrict__ loss, int accumulate) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float p = prob[b];
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    acc += -(target * lp + (1.f - target) * lq);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = weight * sh[0] / (float)B;
    loss[0] = accumulate ? loss[0] + v : v;
  }
}
__global__ void bce_bwd_kernel(const float* __restrict__ prob, float target, int B, float weight,
                               const float* __restrict__ gout, float* __restrict__ dprob) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float p = prob[b];
  const float den = fmaxf((1.f - p) * p, 1e-12f);
  dprob[b] = weight * gout[0] * (p - target) / den / (float)B;
}

// all BCE terms of one discriminator update: H heads x G stacked batches (pointers passed by value)
struct MultiPtr { const float* p[4]; float* d[4]; };
__global__ __launch_bounds__(256) void bce_multi_fwd_kernel(MultiPtr mp, const float* __restrict__ target,
                                                            const float* __restrict__ weight, int G, int H, int B,
                                                            float* __restrict__ loss) {
  __shared__ float sh[256];
  float acc = 0.f;
  const int n = G * H * B;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int b = e % B, gh = e / B, h = gh % H, g = gh / H;
    const float p = mp.p[h][g * B + b];
    const float t = target[g * H + h];
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    acc += weight[g * H + h] * -(t * lp + (1.f - t) * lq);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = sh[0] / (float)B;
}
__global__ void bce_multi_bwd_kernel(MultiPtr mp, const float* __restrict__ target, const float* __restrict__ weight,
                                     int G, int H, int B, const float* __restrict__ gout) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= G * H * B) return;
  const int b = e % B, gh = e / B, h = gh % H, g = gh / H;
  const float p = mp.p[h][g * B + b];
  const float den = fmaxf((1.f - p) * p, 1e-12f);
  mp.d[h][g * B + b] = weight[g * H + h] * gout[0] * (p - target[g * H + h]) / den / (float)B;
}

// ---- class-aware loss ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cal_loss_kernel(const float* __restrict__ S, const int* __restrict__ labels,
                                                       int B, int D, float* __restrict__ loss, int accumulate,
                                                       float* __restrict__ dS) {
  __shared__ float sh[3][256];
  float all = 0.f, pair = 0.f, cnt = 0.f;
  for (int e = threadIdx.x; e < B * B; e += 256) {
    const int i = e / B, j = e - i * B;
    const float v = S[e];
    all += v;
    if (i != j && labels[i] == labels[j]) { pair += v; cnt += 1.f; }
  }
  sh[0][threadIdx.x] = all; sh[1][threadIdx.x] = pair; sh[2][threadIdx.x] = cnt;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
      sh[2][threadIdx.x] += sh[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  const float n = sh[2][0];
  const float diff = n > 0.f ? sh[0][0] / (float)(B * B) - sh[1][0] / n : 0.f;
  const bool active = n > 0.f && diff > 0.f;
  if (threadIdx.x == 0) {
    const float v = active ? diff / (float)D : 0.f;
    loss[0] = accumulate ? loss[0] + v : v;
  }
  if (dS) {
    // d loss / d S, symmetrised so that dX = dS_sym * X
    for (int e = threadIdx.x; e < B * B; e += 256) {
      const int i = e / B, j = e - i * B;
      float g = 0.f;
      if (active) {
        const float m = (i != j && labels[i] == labels[j]) ? 1.f : 0.f;
        g = 2.f * (1.f / (float)(B * B) - m / n) / (float)D;
      }
      dS[e] = g;
    }
  }
}

// ---- speech-encoder front-end -------------------------------------------------------------------------
__global__ void maxpool_w3s2_kernel(const float* __restrict__ x, int W, int C, long long total, float* __restrict__ y) {
  const int Q = C / 4, Wo = W / 2;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(e % Q);
    const long long r = e / Q;         // (b*H + h) * Wo + ox
    const int ox = (int)(r % Wo);
    const long long bh = r / Wo;
    const float* row = x + bh * W * C + q * 4;
    const int x0 = 2 * ox - 1;
    f32x4 m = ld4(row + (long long)(x0 + 1) * C);  // centre tap is always in bounds
    if (x0 >= 0) { const f32x4 v = ld4(row + (long long)x0 * C); for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]); }
    if (x0 + 2 < W) { const f32x4 v = ld4(row + (long long)(x0 + 2) * C); for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]); }
    st4(y + r * C + q * 4, m);
  }
}
__global__ void lstm_cell_kernel(const float* __restrict__ xproj, int ldx, const float* __restrict__ hproj,
                                 const int* __restrict__ lens, int B, int T, int Hd, int step, int reverse,
                                 float* __restrict__ h, float* __restrict__ c, float* __restrict__ out, int ldo) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * Hd) return;
  const int b = e / Hd, j = e - b * Hd;
  const int len = lens[b];
  if (step >= len) return;
  const int t = reverse ? len - 1 - step : step;
  const float* xp = xproj + ((size_t)b * T + t) * ldx;
  const float* hp = hproj + (size_t)b * 4 * Hd;
  const float gi = sigmoidf_(xp[j] + hp[j]);
  const float gf = sigmoidf_(xp[Hd + j] + hp[Hd + j]);
  const float gg = tanhf(xp[2 * Hd + j] + hp[2 * Hd + j]);
  const float go = sigmoidf_(xp[3 * Hd + j] + hp[3 * Hd + j]);
  const float cn = gf * c[e] + gi * gg;
  const float hn = go * tanhf(cn);
  c[e] = cn;
  h[e] = hn;
  out[((size_t)b * T + t) * ldo + j] = hn;
}
// One LSTM time step for every direction in ONE launch: recurrent projection h . W_hh^T fused with the cell update
// (the two kernels above need 3 launches per step and direction).  Block = 8 hidden units x 32 batch slots; h of the
// previous step sits in LDS (rows padded by 4 floats), W_hh is read in its original (4H, H) row-major layout with
// float4 loads along k -- the 8 unit lanes of a wave read 8 rows, the batch lanes share them.
__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ xproj, int ldx,
                                                        const float* __restrict__ whh0, const float* __restrict__ whh1,
                                                        const int* __restrict__ lens, int B, int T, int Hd, int step,
                                                        const float* __restrict__ h_in, float* __restrict__ h_out,
                                                        float* __restrict__ c, float* __restrict__ out, int ldo) {
  extern __shared__ __attribute__((aligned(16))) float hs[];  // h: [32][Hd + 4], then W slice: [4 gates x 8 units][Hd + 4]
  const int d = blockIdx.y;
  const int tid = threadIdx.x;
  const int ul = tid & 7, u = blockIdx.x * 8 + ul, b = tid >> 3;
  const int LDH = Hd + 4, Q = Hd / 4;
  float* wsm = hs + 32 * LDH;
  const float* hin = h_in + (size_t)d * B * Hd;
  const float* w = d ? whh1 : whh0;
  // both tiles with coalesced, independent float4 loads (one memory latency for the whole step)
  for (int e = tid; e < 32 * Q; e += 256) {
    const int r = e / Q, q = e - r * Q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < B) v = *reinterpret_cast<const f32x4*>(hin + (size_t)r * Hd + q * 4);
    *reinterpret_cast<f32x4*>(hs + r * LDH + q * 4) = v;
    const int grow = (r >> 3) * Hd + blockIdx.x * 8 + (r & 7);  // row r = gate * 8 + unit
    *reinterpret_cast<f32x4*>(wsm + r * LDH + q * 4) = *reinterpret_cast<const f32x4*>(w + (size_t)grow * Hd + q * 4);
  }
  __syncthreads();
  const float* w0 = wsm + (0 * 8 + ul) * LDH;
  const float* w1 = wsm + (1 * 8 + ul) * LDH;
  const float* w2 = wsm + (2 * 8 + ul) * LDH;
  const float* w3 = wsm + (3 * 8 + ul) * LDH;
  const float* hp = hs + b * LDH;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll 4
  for (int k = 0; k < Hd; k += 4) {
    const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + k);
    a0 += hv * *reinterpret_cast<const f32x4*>(w0 + k);
    a1 += hv * *reinterpret_cast<const f32x4*>(w1 + k);
    a2 += hv * *reinterpret_cast<const f32x4*>(w2 + k);
    a3 += hv * *reinterpret_cast<const f32x4*>(w3 + k);
  }
  if (b >= B) return;
  const size_t e = ((size_t)d * B + b) * Hd + u;
  const int len = lens[b];
  if (step >= len) {  // finished sequence: the state is carried unchanged (packed-sequence rule)
    h_out[e] = h_in[e];
    return;
  }
  const int t = d ? len - 1 - step : step;
  const float* xp = xproj + ((size_t)b * T + t) * ldx + (size_t)d * 4 * Hd;
  const float gi = sigmoidf_(xp[u] + (a0[0] + a0[1] + a0[2] + a0[3]));
  const float gf = sigmoidf_(xp[Hd + u] + (a1[0] + a1[1] + a1[2] + a1[3]));
  const float gg = tanhf(xp[2 * Hd + u] + (a2[0] + a2[1] + a2[2] + a2[3]));
  const float go = sigmoidf_(xp[3 * Hd + u] + (a3[0] + a3[1] + a3[2] + a3[3]));
  const float cn = gf * c[e] + gi * gg;
  const float hn = go * tanhf(cn);
  c[e] = cn;
  h_out[e] = hn;
  out[((size_t)b * T + t) * ldo + (size_t)d * Hd + u] = hn;
}
__global__ void time_mean_kernel(const float* __restrict__ x, int B, int T, int C, float* __restrict__ y) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int b = e / C, ch = e - b * C;
  float acc = 0.f;
  for (int t = 0; t < T; ++t) acc += x[((size_t)b * T + t) * C + ch];
  y[e] = acc / (float)T;
}

// ---- optimiser ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n4,
                                                   long long n, float lr, float b1, float b2, float eps, int step,
                                                   const int* __restrict__ step_dev, float gscale) {
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const int t = step_dev ? step_dev[0] : step;
    bc[0] = (float)(1.0 - pow((double)b1, (double)t));
    bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
  }
  __syncthreads();
  const float step_size = lr / bc[0];
  const float bc2s = bc[1];
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4;
       e += (long long)gridDim.x * blockDim.x) {
    if (e * 4 + 3 < n) {
      f32x4 pv = ld4(p + e * 4), gv = ld4(g + e * 4), mv = ld4(m + e * 4), vv = ld4(v + e * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gg = gv[j] * gscale;
        mv[j] = b1 * mv[j] + (1.f - b1) * gg;
        vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
        pv[j] -= step_size * mv[j] / (sqrtf(vv[j]) / bc2s + eps);
      }
      st4(p + e * 4, pv); st4(m + e * 4, mv); st4(v + e * 4, vv);
    } else {
      for (long long k = e * 4; k < n; ++k) {
        const float gg = g[k] * gscale;
        const float mm = b1 * m[k] + (1.f - b1) * gg;
        const float vv = b2 * v[k] + (1.f - b2) * gg * gg;
        m[k] = mm; v[k] = vv;
        p[k] -= step_size * mm / (sqrtf(vv) / bc2s + eps);
      }
    }
  }
}
__global__ void ema_kernel(float* __restrict__ avg, const float* __restrict__ p, long long n, float decay) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x)
    avg[e] = decay * avg[e] + (1.f - decay) * p[e];
}
__global__ void axpby_kernel(float* __restrict__ y, const float* __restrict__ x, long long n, float a, float b) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x)
    y[e] = a * x[e] + (b != 0.f ? b * y[e] : 0.f);
}
__global__ void scale_dev_kernel(float* __restrict__ y, const float* __restrict__ x, long long n,
                                 const float* __restrict__ a) {
  const float av = a[0];
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x)
    y[e] = x[e] * av;
}
__global__ void increment_kernel(int* c) { c[0] += 1; }

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long long n4) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x)
    st4(dst + e * 4, ld4(src + e * 4));
}

// ---- row-tiled BatchNorm / activation passes (round 2) --------------------------------------------------------------------
// The grid-stride forms above walk (row, channel quad) pairs: one 64-bit division, up to six coefficient loads and ONE
// 8- or 16-byte activation load in flight per thread and iteration.  Measured (tools/elementwise_bench.py, config 4 shapes):
// bf16 tensors moved at the same ROWS per second as fp32 ones, i.e. at half the bytes per second (1.8 - 3.6 TB/s in the
// backward passes), and the GLU forms at half of that again (both halves' threads load both halves and both compute the
// sigmoid).  These kernels fix a thread to V channels (8 for bf16: 16-byte loads; 4 for fp32) and let it walk rows: every
// coefficient lives in registers, there is no division, two rows' loads are issued before the first is used, and a GLU pair
// (value channel c, gate channel C/2 + c) is ONE thread's work.
// Row-tiled FORWARD kernel for bf16 tensors (measured: wins there; row-tiled backward forms lost to the walkers above and were removed).
typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
template <int V> struct fv { f32x4 v[V / 4]; };

template <int V> __device__ __forceinline__ fv<V> ldv(const float* p) {
  fv<V> r;
#pragma unroll
  for (int k = 0; k < V / 4; ++k) r.v[k] = *reinterpret_cast<const f32x4*>(p + 4 * k);
  return r;
}
template <int V> __device__ __forceinline__ fv<V> ldv(const bf16_t* p) {
  fv<V> r;
  if constexpr (V == 4) {
    r.v[0] = ld4(p);
  } else {
    const u32x4e h = *reinterpret_cast<const u32x4e*>(p);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      r.v[k] = f32x4{__builtin_bit_cast(float, h[2 * k] << 16), __builtin_bit_cast(float, h[2 * k] & 0xffff0000u),
                     __builtin_bit_cast(float, h[2 * k + 1] << 16), __builtin_bit_cast(float, h[2 * k + 1] & 0xffff0000u)};
  }
  return r;
}
template <int V> __device__ __forceinline__ void stv(float* p, const fv<V>& a) {
#pragma unroll
  for (int k = 0; k < V / 4; ++k) *reinterpret_cast<f32x4*>(p + 4 * k) = a.v[k];
}
template <int V> __device__ __forceinline__ void stv(bf16_t* p, const fv<V>& a) {
  if constexpr (V == 4) {
    st4(p, a.v[0]);
  } else {
    u32x4e h;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x2_ lo = {a.v[k][0], a.v[k][1]}, hi = {a.v[k][2], a.v[k][3]};
      h[2 * k] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2_));
      h[2 * k + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2_));
    }
    *reinterpret_cast<u32x4e*>(p) = h;
  }
}

// rows of this block: the rows are `G` independent BatchNorm batches of Rg rows, each walked by ppg blocks
struct RowSpan { long long r0, r1; int grp; };
__device__ __forceinline__ RowSpan row_span(long long M, int ppg, long long Rg) {
  RowSpan s;
  s.grp = blockIdx.x / ppg;
  const int pp = blockIdx.x - s.grp * ppg;
  const long long chunk = (Rg + ppg - 1) / ppg;
  s.r0 = s.grp * Rg + pp * chunk;
  const long long gend = (s.grp + 1) * Rg < M ? (s.grp + 1) * Rg : M;
  s.r1 = s.r0 + chunk < gend ? s.r0 + chunk : gend;
  return s;
}

template <typename T, int V, int ACT = -1>
__global__ __launch_bounds__(256) void bn_act_fwd_rows_kernel(const T* __restrict__ y, long long M, int C,
                                                              const float* __restrict__ coef0, int act_rt,
                                                              const T* __restrict__ residual, T* __restrict__ out,
                                                              int lgc, int ppg, long long Rg) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int cpb = 1 << lgc, rpb = 256 >> lgc;
  const int ql = threadIdx.x & (cpb - 1), rl = threadIdx.x >> lgc;
  const bool glu = act == S2I_ACT_GLU;
  const int Cout = glu ? C / 2 : C;
  const int c0 = (blockIdx.y * cpb + ql) * V;
  if (c0 >= Cout) return;
  const RowSpan sp = row_span(M, ppg, Rg);
  const float* scale = coef0 + (size_t)sp.grp * 4 * C + 2 * C;
  const float* shift = scale + C;
  const fv<V> sa = ldv<V>(scale + c0), ta = ldv<V>(shift + c0);
  fv<V> sg, tg;
  if (glu) { sg = ldv<V>(scale + Cout + c0); tg = ldv<V>(shift + Cout + c0); }
  auto one = [&](long long row, const fv<V>& ya, const fv<V>& yx) {   // yx: gate half (GLU) or residual
    fv<V> o;
#pragma unroll
    for (int k = 0; k < V / 4; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z = sa.v[k][j] * ya.v[k][j] + ta.v[k][j];
        if (glu) z *= sigmoid_gate_<T>(sg.v[k][j] * yx.v[k][j] + tg.v[k][j]);
        else if (act == S2I_ACT_LRELU) z = z > 0.f ? z : 0.2f * z;
        if (!glu && residual) z += yx.v[k][j];
        o.v[k][j] = z;
      }
    stv<V>(out + row * Cout + c0, o);
  };
  const bool two = glu || residual != nullptr;
  const T* second = glu ? y + Cout : residual;
  const long long ld2 = glu ? C : Cout;
  long long row = sp.r0 + rl;
  for (; row + rpb < sp.r1; row += 2 * rpb) {
    const fv<V> a0 = ldv<V>(y + row * C + c0), a1 = ldv<V>(y + (row + rpb) * C + c0);
    fv<V> x0 = a0, x1 = a1;
    if (two) { x0 = ldv<V>(second + row * ld2 + c0); x1 = ldv<V>(second + (row + rpb) * ld2 + c0); }
    one(row, a0, x0);
    one(row + rpb, a1, x1);
  }
  if (row < sp.r1) {
    const fv<V> a0 = ldv<V>(y + row * C + c0);
    fv<V> x0 = a0;
    if (two) x0 = ldv<V>(second + row * ld2 + c0);
    one(row, a0, x0);
  }
}

// host geometry of the row-tiled kernels: threads across the channel vectors (a power of two), the rest of the block
// down the rows; enough blocks along the rows to keep ~16 waves per CU busy with at least a few trips each
// blocks along the rows of one BatchNorm group: `want` rows per thread where the tensor is large, but never so few blocks
// that the chip is under-filled (small tensors: down to one row per thread -- a short kernel is all latency, and a
// thread that walks 8 rows one pair at a time takes four memory round trips where one would do), at most 4096 blocks
static int rows_ppg(long long Rg, int rpb, int groups, int gy, int want) {
  const long long per = (long long)groups * gy;
  long long ppg = (Rg + (long long)rpb * want - 1) / ((long long)rpb * want);
  const long long fill = (2048 + per - 1) / per;
  if (ppg < fill) ppg = fill;
  const long long most = (Rg + rpb - 1) / rpb;               // one row per thread
  if (ppg > most) ppg = most;
  const long long cap = 4096 / per > 0 ? 4096 / per : 1;
  if (ppg > cap) ppg = cap;
  if (ppg < 1) ppg = 1;
  return (int)ppg;
}
struct RowGeom { int lgc, gy, ppg; };
static RowGeom row_geom(int nvec, long long Rg, int groups, int want_parts) {
  RowGeom g;
  g.lgc = 0;
  while ((1 << g.lgc) < nvec && g.lgc < 8) ++g.lgc;
  const int cpb = 1 << g.lgc, rpb = 256 / cpb;
  g.gy = (nvec + cpb - 1) / cpb;
  if (want_parts > 0) { g.ppg = want_parts; return g; }
  g.ppg = rows_ppg(Rg, rpb, groups, g.gy, 8);
  return g;
}

inline int grid_for(long long total, int block = 256, int cap = 2048 * 4) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int s2i_colstats(const float* y, long long M, int C, int ldy, float* part, int nparts, void* stream) {
  S2I_REQUIRE(y && part && M > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && nparts > 0, "colstats: bad args");
  RedGeom g = red_geom(C);
  hipLaunchKernelGGL((colreduce_kernel<0, float>), dim3(nparts, g.gy), dim3(256), 0, ST, y, ldy, (const float*)nullptr, 0,
                     M, C, (const float*)nullptr, 0, part, nparts, g.cpb, nparts, M);
  S2I_LAUNCH_CHECK("colstats");
  return 0;
}

static int launch_finalize(int mode, const float* part, int nparts, int groups, int C, long long count,
                           const float* gamma, const float* beta, float* rmean, float* rvar, float momentum, float eps,
                           float* out, float* dgamma, float* dbeta, int accumulate, void* stream,
                           long long* nbt = nullptr) {
  S2I_REQUIRE(part && out && nparts > 0 && C > 0 && C % 4 == 0 && count > 0, "bn finalize: bad args");
  S2I_REQUIRE(groups >= 1 && nparts % groups == 0, "bn finalize: %d partial rows do not split into %d groups", nparts,
              groups);
  const int ppg = nparts / groups;
  if (ppg <= 8) {
    const int grid = (C + 255) / 256;
    if (mode == 0)
      hipLaunchKernelGGL((bn_finalize_small_kernel<0>), dim3(grid), dim3(256), 0, ST, part, ppg, groups, C, (double)count,
                         gamma, beta, rmean, rvar, momentum, eps, out, dgamma, dbeta, accumulate, nbt);
    else
      hipLaunchKernelGGL((bn_finalize_small_kernel<1>), dim3(grid), dim3(256), 0, ST, part, ppg, groups, C, (double)count,
                         gamma, beta, rmean, rvar, momentum, eps, out, dgamma, dbeta, accumulate, nbt);
    S2I_LAUNCH_CHECK("bn_finalize_small");
    return 0;
  }
  const int Q = C / 4;
  // quads per block: few for narrow layers (their partial lists are the long ones), up to 32 for wide layers
  int qpb = 1;
  while (qpb < 32 && qpb * 32 < Q) qpb <<= 1;
  const int grid = (Q + qpb - 1) / qpb;
  // row lanes per group: a power of two, no more than the list is long, groups side by side in at most 256 threads.  (1024
  // threads and 64 KB of LDS until round 3: such a block cannot start on a CU that runs three matrix blocks of another stream
  // -- 123 KB of LDS, 12 of 16 wave slots -- and waited for the tail of that kernel: 44 us per finalize inside the step against
  // 10 us alone.  A 256-thread block with 16 KB fits beside them.)
  const int cap = s2i_tune(S2I_TUNE_FINALIZE_THREADS, 256);
  int lpg = 1;
  while (lpg < ppg && lpg * 2 * groups * qpb <= cap) lpg <<= 1;
  int nthreads = qpb * lpg * groups;
  nthreads = (nthreads + 63) & ~63;
  if (nthreads > 1024) nthreads = 1024;
  const int nwaves = nthreads / 64;
  size_t slots = (size_t)nthreads > (size_t)nwaves * qpb ? (size_t)nthreads : (size_t)nwaves * qpb;
  if (slots < (size_t)groups * qpb) slots = (size_t)groups * qpb;
  const size_t shbytes = slots * 8 * sizeof(double);
  if (mode == 0)
    hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(grid), dim3(nthreads), shbytes, ST, part, ppg, groups, C, (double)count,
                       gamma, beta, rmean, rvar, momentum, eps, out, dgamma, dbeta, accumulate, qpb, lpg, nbt);
  else
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(grid), dim3(nthreads), shbytes, ST, part, ppg, groups, C, (double)count,
                       gamma, beta, rmean, rvar, momentum, eps, out, dgamma, dbeta, accumulate, qpb, lpg, nbt);
  S2I_LAUNCH_CHECK("bn_finalize");
  return 0;
}

extern "C" int s2i_bn_finalize(const float* part, int nparts, int groups, int C, long long count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var,
                               long long* num_batches_tracked, float momentum, float eps, float* out4, void* stream) {
  S2I_REQUIRE(gamma && beta, "bn_finalize: null affine parameters");
  S2I_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
  return launch_finalize(0, part, nparts, groups, C, count, gamma, beta, running_mean, running_var, momentum, eps,
                         out4, nullptr, nullptr, 0, stream, num_batches_tracked);
}

extern "C" int s2i_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* out4, void* stream) {
  S2I_REQUIRE(C > 0 && gamma && beta && running_mean && running_var && out4, "bn_eval_coeffs: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, C, gamma, beta, running_mean,
                     running_var, eps, out4);
  S2I_LAUNCH_CHECK("bn_eval_coeffs");
  return 0;
}

#define S2I_DT_CHECK(dt, name) S2I_REQUIRE((dt) == S2I_DT_F32 || (dt) == S2I_DT_BF16, name ": unknown dtype %d", (dt))

template <typename T>
static int bn_act_forward_impl(const T* y, long long M, int groups, int C, const float* coef4, int act,
                               const T* residual, T* out, void* stream) {
  S2I_REQUIRE(y && coef4 && out && M > 0 && C > 0, "bn_act_forward: bad args");
  S2I_REQUIRE(groups >= 1 && M % groups == 0 && M < (1ll << 31), "bn_act_forward: rows do not split into groups");
  S2I_REQUIRE(act == S2I_ACT_GLU ? C % 8 == 0 : C % 4 == 0, "bn_act_forward: C=%d not aligned for act %d", C, act);
  S2I_REQUIRE(!(residual && act == S2I_ACT_GLU), "bn_act_forward: residual with GLU unsupported");
  const int Cout = act == S2I_ACT_GLU ? C / 2 : C;
  constexpr bool is16 = sizeof(T) == 2;
  if (is16) {   // fp32 tensors: no gain from the row-tiled form
#define S2I_FWDR(VV, ACTV) hipLaunchKernelGGL((bn_act_fwd_rows_kernel<T, VV, ACTV>), dim3(groups * g.ppg, g.gy), dim3(256), 0, ST, \
                                              y, M, C, coef4, act, residual, out, g.lgc, g.ppg, M / groups)
    if (is16 && (Cout % 8) == 0) {
      const RowGeom g = row_geom(Cout / 8, M / groups, groups, 0);
      if (act == S2I_ACT_GLU) S2I_FWDR(8, S2I_ACT_GLU);
      else if (act == S2I_ACT_LRELU) S2I_FWDR(8, S2I_ACT_LRELU);
      else if (act == S2I_ACT_NONE) S2I_FWDR(8, S2I_ACT_NONE);
      else S2I_FWDR(8, -1);
    } else {
      const RowGeom g = row_geom(Cout / 4, M / groups, groups, 0);
      S2I_FWDR(4, -1);
    }
#undef S2I_FWDR
    S2I_LAUNCH_CHECK("bn_act_forward(rows)");
    return 0;
  }
  const long long total = M * (Cout / 4);
#define S2I_FWD(ACTV) hipLaunchKernelGGL((bn_act_fwd_kernel<T, ACTV>), dim3(grid_for(total)), dim3(256), 0, ST, y, M, C, coef4, \
                                         act, residual, out, groups, (unsigned)(M / groups))
  if (act == S2I_ACT_GLU) S2I_FWD(S2I_ACT_GLU);
  else if (act == S2I_ACT_LRELU) S2I_FWD(S2I_ACT_LRELU);
  else if (act == S2I_ACT_NONE) S2I_FWD(S2I_ACT_NONE);
  else S2I_FWD(-1);
#undef S2I_FWD
  S2I_LAUNCH_CHECK("bn_act_forward");
  return 0;
}
extern "C" int s2i_bn_act_forward(const float* y, long long M, int groups, int C, const float* coef4, int act,
                                  const float* residual, float* out, void* stream) {
  return bn_act_forward_impl<float>(y, M, groups, C, coef4, act, residual, out, stream);
}
extern "C" int s2i_bn_act_forward_dt(int dtype, const void* y, long long M, int groups, int C, const float* coef4, int act,
                                     const void* residual, void* out, void* stream) {
  S2I_DT_CHECK(dtype, "bn_act_forward");
  if (dtype == S2I_DT_BF16)
    return bn_act_forward_impl<bf16_t>((const bf16_t*)y, M, groups, C, coef4, act, (const bf16_t*)residual, (bf16_t*)out, stream);
  return bn_act_forward_impl<float>((const float*)y, M, groups, C, coef4, act, (const float*)residual, (float*)out, stream);
}

template <typename T>
static int bn_act_bwd_reduce_impl(const T* y, const T* dout, int lddout, long long M, int groups, int C,
                                  const float* coef4, int act, float* part, int nparts, void* stream) {
  S2I_REQUIRE(y && dout && coef4 && part && M > 0 && nparts > 0, "bn_act_bwd_reduce: bad args");
  S2I_REQUIRE(groups >= 1 && M % groups == 0 && nparts % groups == 0, "bn_act_bwd_reduce: bad grouping");
  S2I_REQUIRE(act == S2I_ACT_GLU ? C % 8 == 0 : C % 4 == 0, "bn_act_bwd_reduce: C alignment");
  S2I_REQUIRE(lddout % 4 == 0, "bn_act_bwd_reduce: lddout alignment");
  RedGeom g = red_geom(C);
  // the activation as a template constant: see colreduce_kernel
  const bool spec = true;
#define S2I_RED(ACTV) hipLaunchKernelGGL((colreduce_kernel<1, T, ACTV, 4>), dim3(nparts, g.gy), dim3(256), 0, ST, y, C, \
                                         dout, lddout, M, C, coef4, act, part, nparts, g.cpb, nparts / groups, M / groups)
  if (spec && act == S2I_ACT_LRELU) S2I_RED(S2I_ACT_LRELU);
  else if (spec && act == S2I_ACT_GLU) S2I_RED(S2I_ACT_GLU);
  else if (spec && act == S2I_ACT_NONE) S2I_RED(S2I_ACT_NONE);
  else S2I_RED(-1);
#undef S2I_RED
  S2I_LAUNCH_CHECK("bn_act_bwd_reduce");
  return 0;
}
extern "C" int s2i_bn_act_bwd_reduce(const float* y, const float* dout, int lddout, long long M, int groups, int C,
                                     const float* coef4, int act, float* part, int nparts, void* stream) {
  return bn_act_bwd_reduce_impl<float>(y, dout, lddout, M, groups, C, coef4, act, part, nparts, stream);
}
extern "C" int s2i_bn_act_bwd_reduce_dt(int dtype, const void* y, const void* dout, int lddout, long long M, int groups,
                                        int C, const float* coef4, int act, float* part, int nparts, void* stream) {
  S2I_DT_CHECK(dtype, "bn_act_bwd_reduce");
  if (dtype == S2I_DT_BF16)
    return bn_act_bwd_reduce_impl<bf16_t>((const bf16_t*)y, (const bf16_t*)dout, lddout, M, groups, C, coef4, act, part, nparts, stream);
  return bn_act_bwd_reduce_impl<float>((const float*)y, (const float*)dout, lddout, M, groups, C, coef4, act, part, nparts, stream);
}

extern "C" int s2i_bn_bwd_finalize(const float* part, int nparts, int groups, int C, long long count, float* dgamma,
                                   float* dbeta, int accumulate, float* red2, void* stream) {
  return launch_finalize(1, part, nparts, groups, C, count, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, red2, dgamma,
                         dbeta, accumulate, stream);
}

template <typename T>
static int bn_act_bwd_apply_impl(const T* y, const T* dout, int lddout, long long M, int groups, int C,
                                 const float* coef4, const float* red2, int act, T* dy, void* stream) {
  S2I_REQUIRE(y && dout && coef4 && red2 && dy && M > 0, "bn_act_bwd_apply: bad args");
  S2I_REQUIRE(groups >= 1 && M % groups == 0 && M < (1ll << 31), "bn_act_bwd_apply: rows do not split into groups");
  S2I_REQUIRE(act == S2I_ACT_GLU ? C % 8 == 0 : C % 4 == 0, "bn_act_bwd_apply: C alignment");
  S2I_REQUIRE(lddout % 4 == 0, "bn_act_bwd_apply: lddout alignment");
  if (act == S2I_ACT_GLU || act == S2I_ACT_LRELU || act == S2I_ACT_NONE) {
    RedGeom g = red_geom(C);
    const long long Rg = M / groups;
    const int rpb = 256 / g.cpb;
    const int ppg = rows_ppg(Rg, rpb, groups, g.gy, 16);
#define S2I_APPW(ACTV) hipLaunchKernelGGL((bn_act_bwd_apply_walk_kernel<T, ACTV>), dim3(groups * ppg, g.gy), dim3(256), 0, ST, y, \
                                          dout, lddout, M, C, coef4, red2, dy, g.cpb, ppg, Rg)
    if (act == S2I_ACT_GLU) S2I_APPW(S2I_ACT_GLU);
    else if (act == S2I_ACT_LRELU) S2I_APPW(S2I_ACT_LRELU);
    else S2I_APPW(S2I_ACT_NONE);
#undef S2I_APPW
    S2I_LAUNCH_CHECK("bn_act_bwd_apply(walk)");
    return 0;
  }
#define S2I_APP(ACTV) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T, ACTV>), dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, y, dout, \
                                         lddout, M, C, coef4, red2, act, dy, groups, (unsigned)(M / groups))
  if (act == S2I_ACT_GLU) S2I_APP(S2I_ACT_GLU);
  else if (act == S2I_ACT_LRELU) S2I_APP(S2I_ACT_LRELU);
  else if (act == S2I_ACT_NONE) S2I_APP(S2I_ACT_NONE);
  else S2I_APP(-1);
#undef S2I_APP
  S2I_LAUNCH_CHECK("bn_act_bwd_apply");
  return 0;
}
extern "C" int s2i_bn_act_bwd_apply(const float* y, const float* dout, int lddout, long long M, int groups, int C,
                                    const float* coef4, const float* red2, int act, float* dy, void* stream) {
  return bn_act_bwd_apply_impl<float>(y, dout, lddout, M, groups, C, coef4, red2, act, dy, stream);
}
extern "C" int s2i_bn_act_bwd_apply_dt(int dtype, const void* y, const void* dout, int lddout, long long M, int groups,
                                       int C, const float* coef4, const float* red2, int act, void* dy, void* stream) {
  S2I_DT_CHECK(dtype, "bn_act_bwd_apply");
  if (dtype == S2I_DT_BF16)
    return bn_act_bwd_apply_impl<bf16_t>((const bf16_t*)y, (const bf16_t*)dout, lddout, M, groups, C, coef4, red2, act, (bf16_t*)dy, stream);
  return bn_act_bwd_apply_impl<float>((const float*)y, (const float*)dout, lddout, M, groups, C, coef4, red2, act, (float*)dy, stream);
}

template <typename T>
static int act_backward_impl(const T* out, const T* dout, int lddout, long long M, int C, int act, T* dy, void* stream) {
  S2I_REQUIRE(out && dout && dy && M > 0 && C > 0 && C % 4 == 0 && lddout % 4 == 0, "act_backward: bad args");
  if (act == S2I_ACT_LRELU)
    hipLaunchKernelGGL((act_bwd_kernel<T, S2I_ACT_LRELU>), dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, out, dout, lddout, M, C, act, dy);
  else if (act == S2I_ACT_TANH)
    hipLaunchKernelGGL((act_bwd_kernel<T, S2I_ACT_TANH>), dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, out, dout, lddout, M, C, act, dy);
  else
    hipLaunchKernelGGL((act_bwd_kernel<T, -1>), dim3(grid_for(M * (C / 4))), dim3(256), 0, ST, out, dout, lddout, M, C, act, dy);
  S2I_LAUNCH_CHECK("act_backward");
  return 0;
}
extern "C" int s2i_act_backward(const float* out, const float* dout, int lddout, long long M, int C, int act,
                                float* dy, void* stream) {
  return act_backward_impl<float>(out, dout, lddout, M, C, act, dy, stream);
}
extern "C" int s2i_act_backward_dt(int dtype, const void* out, const void* dout, int lddout, long long M, int C, int act,
                                   void* dy, void* stream) {
  S2I_DT_CHECK(dtype, "act_backward");
  if (dtype == S2I_DT_BF16)
    return act_backward_impl<bf16_t>((const bf16_t*)out, (const bf16_t*)dout, lddout, M, C, act, (bf16_t*)dy, stream);
  return act_backward_impl<float>((const float*)out, (const float*)dout, lddout, M, C, act, (float*)dy, stream);
}

extern "C" int s2i_glu_forward(const float* x, long long M, int C, float* out, void* stream) {
  S2I_REQUIRE(x && out && M > 0 && C > 0 && C % 2 == 0, "glu_forward: bad args");
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(M * (C / 2))), dim3(256), 0, ST, x, M, C, out);
  S2I_LAUNCH_CHECK("glu_forward");
  return 0;
}
extern "C" int s2i_glu_backward(const float* x, const float* dout, long long M, int C, float* dx, void* stream) {
  S2I_REQUIRE(x && dout && dx && M > 0 && C > 0 && C % 2 == 0, "glu_backward: bad args");
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(M * (C / 2))), dim3(256), 0, ST, x, dout, M, C, dx);
  S2I_LAUNCH_CHECK("glu_backward");
  return 0;
}

extern "C" int s2i_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream) {
  S2I_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nchw_to_nhwc: bad args");
  if (C == 3 && Cp == 4) {
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3(grid_for((long long)B * H * W)), dim3(256), 0, ST, src, dst, B,
                       H * W);
  } else {
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for((long long)B * H * W * Cp)), dim3(256), 0, ST, src, dst, B,
                       C, H * W, Cp);
  }
  S2I_LAUNCH_CHECK("nchw_to_nhwc");
  return 0;
}
extern "C" int s2i_nhwc_to_nchw(const float* src, int lds, float* dst, int B, int C, int H, int W, void* stream) {
  S2I_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && lds >= C, "nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, ST, src, lds, dst,
                     B, C, H * W);
  S2I_LAUNCH_CHECK("nhwc_to_nchw");
  return 0;
}
/* the same with the NHWC side stored as bf16 (the NCHW side stays fp32: the module boundary) */
extern "C" int s2i_nchw_to_nhwc_dt(int dtype, const float* src, void* dst, int B, int C, int H, int W, int Cp, void* stream) {
  S2I_DT_CHECK(dtype, "nchw_to_nhwc");
  if (dtype == S2I_DT_F32) return s2i_nchw_to_nhwc(src, (float*)dst, B, C, H, W, Cp, stream);
  S2I_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && Cp >= C, "nchw_to_nhwc: bad args");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for((long long)B * H * W * Cp)), dim3(256), 0, ST, src,
                     (bf16_t*)dst, B, C, H * W, Cp);
  S2I_LAUNCH_CHECK("nchw_to_nhwc");
  return 0;
}
extern "C" int s2i_nhwc_to_nchw_dt(int dtype, const void* src, int lds, float* dst, int B, int C, int H, int W, void* stream) {
  S2I_DT_CHECK(dtype, "nhwc_to_nchw");
  if (dtype == S2I_DT_F32) return s2i_nhwc_to_nchw((const float*)src, lds, dst, B, C, H, W, stream);
  S2I_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0 && lds >= C, "nhwc_to_nchw: bad args");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for((long long)B * C * H * W)), dim3(256), 0, ST,
                     (const bf16_t*)src, lds, dst, B, C, H * W);
  S2I_LAUNCH_CHECK("nhwc_to_nchw");
  return 0;
}

extern "C" int s2i_image_to_u8(const float* src, int lds, unsigned char* dst, long long npix, void* stream) {
  S2I_REQUIRE(src && dst && lds >= 3 && npix > 0, "image_to_u8: bad args");
  hipLaunchKernelGGL(image_to_u8_kernel, dim3(grid_for(npix)), dim3(256), 0, ST, src, lds, dst, npix);
  S2I_LAUNCH_CHECK("image_to_u8");
  return 0;
}

extern "C" int s2i_u8_to_image(const unsigned char* src, float* dst, int B, int H, int W, void* stream) {
  S2I_REQUIRE(src && dst && B > 0 && H > 0 && W > 0, "u8_to_image: bad args");
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(u8_to_image_kernel, dim3(grid_for(npix)), dim3(256), 0, ST, src, dst, H * W, npix);
  S2I_LAUNCH_CHECK("u8_to_image");
  return 0;
}

static int spatial_segments(int HW) {
  int S = HW / 64;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  return S;
}
extern "C" size_t s2i_spatial_sum_workspace_bytes(int B, int HW, int C) {
  return (size_t)B * spatial_segments(HW) * C * sizeof(float);
}
template <typename T>
static int spatial_sum_impl(const T* src, int ld, int B, int HW, int C, float* dst, void* ws, size_t ws_bytes,
                            void* stream) {
  S2I_REQUIRE(src && dst && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C,
              "spatial_sum: bad args");
  const int S = spatial_segments(HW);
  S2I_REQUIRE(ws && ws_bytes >= (size_t)B * S * C * sizeof(float), "spatial_sum: workspace too small");
  RedGeom g = red_geom(C);
  hipLaunchKernelGGL(spatial_sum_stage1<T>, dim3(B, S, g.gy), dim3(256), 0, ST, src, ld, HW, C, S, (float*)ws, g.cpb);
  S2I_LAUNCH_CHECK("spatial_sum_stage1");
  hipLaunchKernelGGL(spatial_sum_stage2, dim3((B * C + 255) / 256), dim3(256), 0, ST, (const float*)ws, B, S, C, dst);
  S2I_LAUNCH_CHECK("spatial_sum_stage2");
  return 0;
}
extern "C" int s2i_spatial_sum(const float* src, int ld, int B, int HW, int C, float* dst, void* ws, size_t ws_bytes,
                               void* stream) {
  return spatial_sum_impl<float>(src, ld, B, HW, C, dst, ws, ws_bytes, stream);
}
extern "C" int s2i_spatial_sum_dt(int dtype, const void* src, int ld, int B, int HW, int C, float* dst, void* ws,
                                  size_t ws_bytes, void* stream) {
  S2I_DT_CHECK(dtype, "spatial_sum");
  if (dtype == S2I_DT_BF16) return spatial_sum_impl<bf16_t>((const bf16_t*)src, ld, B, HW, C, dst, ws, ws_bytes, stream);
  return spatial_sum_impl<float>((const float*)src, ld, B, HW, C, dst, ws, ws_bytes, stream);
}

extern "C" int s2i_cvec_bias_table(const float* cvec, const float* packed, int B, int Cc, int Ip, int Op, int N,
                                   float* table, void* ws, size_t ws_bytes, void* stream) {
  S2I_REQUIRE(cvec && packed && table && B > 0 && Cc > 0 && Cc <= Ip && N > 0 && N <= Op, "cvec_bias_table: bad args");
  S2I_REQUIRE(ws && ws_bytes >= (size_t)B * 9 * N * sizeof(float), "cvec_bias_table: workspace too small");
  hipLaunchKernelGGL(cvec_tap_table_kernel, dim3(B * 9), dim3(256), Cc * sizeof(float), ST, cvec, packed, Cc, Ip, Op, N,
                     (float*)ws);
  S2I_LAUNCH_CHECK("cvec_tap_table");
  hipLaunchKernelGGL(cvec_bias_table_kernel, dim3((B * 9 * N + 255) / 256), dim3(256), 0, ST, (const float*)ws, B, N,
                     table);
  S2I_LAUNCH_CHECK("cvec_bias_table");
  return 0;
}
static int border_segments(int H) {
  int S = H / 4;
  if (S > 32) S = 32;
  if (S < 1) S = 1;
  return S;
}
extern "C" size_t s2i_border_sums_workspace_bytes(int B, int H, int W, int C) {
  return (size_t)B * border_segments(H) * 9 * C * sizeof(float);
}
template <typename T>
static int tap_sums_impl(const T* dy, int B, int H, int W, int C, float* tapsum, void* ws, size_t ws_bytes, void* stream) {
  S2I_REQUIRE(dy && tapsum && B > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "tap_sums: bad args");
  const int S = border_segments(H);
  S2I_REQUIRE(ws && ws_bytes >= (size_t)B * S * 9 * C * sizeof(float), "tap_sums: workspace too small");
  RedGeom g = red_geom(C);
  hipLaunchKernelGGL(border_sums_stage1<T>, dim3(B, S, g.gy), dim3(256), 0, ST, dy, H, W, C, S, (float*)ws, g.cpb);
  S2I_LAUNCH_CHECK("border_sums_stage1");
  hipLaunchKernelGGL(border_sums_stage2, dim3((B * C + 255) / 256), dim3(256), 0, ST, (const float*)ws, B, S, C,
                     tapsum);
  S2I_LAUNCH_CHECK("border_sums_stage2");
  return 0;
}
extern "C" int s2i_tap_sums(const float* dy, int B, int H, int W, int C, float* tapsum, void* ws, size_t ws_bytes,
                            void* stream) {
  return tap_sums_impl<float>(dy, B, H, W, C, tapsum, ws, ws_bytes, stream);
}
extern "C" int s2i_tap_sums_dt(int dtype, const void* dy, int B, int H, int W, int C, float* tapsum, void* ws,
                               size_t ws_bytes, void* stream) {
  S2I_DT_CHECK(dtype, "tap_sums");
  if (dtype == S2I_DT_BF16) return tap_sums_impl<bf16_t>((const bf16_t*)dy, B, H, W, C, tapsum, ws, ws_bytes, stream);
  return tap_sums_impl<float>((const float*)dy, B, H, W, C, tapsum, ws, ws_bytes, stream);
}

/* element type conversion of a contiguous tensor: dst_dtype[n] = src_dtype[n] (n % 4 == 0) */
extern "C" int s2i_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long long n, void* stream) {
  S2I_DT_CHECK(src_dtype, "cast");
  S2I_DT_CHECK(dst_dtype, "cast");
  S2I_REQUIRE(src && dst && n > 0 && (n % 4) == 0 && src_dtype != dst_dtype, "cast: bad args");
  if (src_dtype == S2I_DT_F32)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(grid_for(n / 4)), dim3(256), 0, ST, (const float*)src, (bf16_t*)dst, n / 4);
  else
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(grid_for(n / 4)), dim3(256), 0, ST, (const bf16_t*)src, (float*)dst, n / 4);
  S2I_LAUNCH_CHECK("cast");
  return 0;
}
extern "C" int s2i_cvec_grads(const float* cvec, const float* packed, const float* tapsum, int B, int Cc, int Ip, int Op,
                              int N, int O, int I_total, float* dc, float* dw_oihw, int accumulate, void* stream) {
  S2I_REQUIRE(cvec && packed && tapsum && B > 0 && Cc > 0 && Cc <= Ip && N > 0 && N <= Op && O <= N && I_total >= Cc,
              "cvec_grads: bad args");
  if (dc) {
    S2I_REQUIRE(N % 4 == 0 && Op % 4 == 0, "cvec_grads: N must be a multiple of 4");
    hipLaunchKernelGGL(cvec_dc_kernel, dim3(B, (Cc + 63) / 64), dim3(256), 0, ST, packed, tapsum, B, Cc, Ip, Op, N, dc);
    S2I_LAUNCH_CHECK("cvec_dc");
  }
  if (dw_oihw) {
    hipLaunchKernelGGL(cvec_dw_kernel, dim3((O * Cc * 9 + 255) / 256), dim3(256), 0, ST, cvec, tapsum, B, Cc, N, O,
                       I_total, dw_oihw, accumulate);
    S2I_LAUNCH_CHECK("cvec_dw");
  }
  return 0;
}

extern "C" int s2i_reparam_forward(const float* h, const float* eps, int B, int E, float* c, void* stream) {
  S2I_REQUIRE(h && eps && c && B > 0 && E > 0, "reparam_forward: bad args");
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3((B * E + 255) / 256), dim3(256), 0, ST, h, eps, B, E, c);
  S2I_LAUNCH_CHECK("reparam_forward");
  return 0;
}
extern "C" int s2i_reparam_backward(const float* h, const float* eps, const float* dc, const float* dmu,
                                    const float* dlogvar, int B, int E, float* dh, void* stream) {
  S2I_REQUIRE(h && eps && dh && B > 0 && E > 0, "reparam_backward: bad args");
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3((B * E + 255) / 256), dim3(256), 0, ST, h, eps, dc, dmu, dlogvar, B, E,
                     dh);
  S2I_LAUNCH_CHECK("reparam_backward");
  return 0;
}
extern "C" int s2i_kl_forward(const float* mu, int ldmu, const float* logvar, int ldlv, int B, int E, float* kl,
                              void* stream) {
  S2I_REQUIRE(mu && logvar && kl && B > 0 && E > 0, "kl_forward: bad args");
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(1), dim3(256), 0, ST, mu, ldmu, logvar, ldlv, B, E, kl);
  S2I_LAUNCH_CHECK("kl_forward");
  return 0;
}
extern "C" int s2i_kl_backward(const float* mu, int ldmu, const float* logvar, int ldlv, int B, int E,
                               const float* gout, float* dmu, float* dlogvar, void* stream) {
  S2I_REQUIRE(mu && logvar && gout && dmu && dlogvar && B > 0 && E > 0, "kl_backward: bad args");
  hipLaunchKernelGGL(kl_bwd_kernel, dim3((B * E + 255) / 256), dim3(256), 0, ST, mu, ldmu, logvar, ldlv, B, E, gout,
                     dmu, dlogvar);
  S2I_LAUNCH_CHECK("kl_backward");
  return 0;
}

extern "C" int s2i_logit_forward(const float* x, const float* w, const float* bias, int B, int C, float* prob,
                                 void* stream) {
  S2I_REQUIRE(x && w && prob && B > 0 && C > 0, "logit_forward: bad args");
  hipLaunchKernelGGL(logit_fwd_kernel, dim3(B), dim3(256), 0, ST, x, w, bias, C, prob);
  S2I_LAUNCH_CHECK("logit_forward");
  return 0;
}
extern "C" int s2i_logit_backward(const float* x, const float* w, const float* prob, const float* dprob, int B, int C,
                                  float* dx, int acc_dx, float* dw, float* dbias, int acc_dw, void* stream) {
  S2I_REQUIRE(x && w && prob && dprob && B > 0 && C > 0, "logit_backward: bad args");
  hipLaunchKernelGGL(logit_bwd_kernel, dim3((16 * C + 255) / 256), dim3(256), 0, ST, x, w, prob, dprob, B, C, dx,
                     acc_dx, dw, dbias, acc_dw);
  S2I_LAUNCH_CHECK("logit_backward");
  return 0;
}
extern "C" int s2i_bce_forward(const float* prob, float target, int B, float weight, float* loss, int accumulate,
                               void* stream) {
  S2I_REQUIRE(prob && loss && B > 0, "bce_forward: bad args");
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(256), 0, ST, prob, target, B, weight, loss, accumulate);
  S2I_LAUNCH_CHECK("bce_forward");
  return 0;
}
extern "C" int s2i_bce_backward(const float* prob, float target, int B, float weight, const float* gout,
                                float* dprob, void* stream) {
  S2I_REQUIRE(prob && gout && dprob && B > 0, "bce_backward: bad args");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, ST, prob, target, B, weight, gout, dprob);
  S2I_LAUNCH_CHECK("bce_backward");
  return 0;
}

extern "C" int s2i_bce_multi_forward(const float* const* probs, const float* target, const float* weight, int G, int H,
                                     int B, float* loss, void* stream) {
  S2I_REQUIRE(probs && target && weight && loss && G > 0 && H > 0 && H <= 4 && B > 0, "bce_multi_forward: bad args");
  MultiPtr mp = {};
  for (int h = 0; h < H; ++h) { S2I_REQUIRE(probs[h], "bce_multi_forward: null head"); mp.p[h] = probs[h]; }
  hipLaunchKernelGGL(bce_multi_fwd_kernel, dim3(1), dim3(256), 0, ST, mp, target, weight, G, H, B, loss);
  S2I_LAUNCH_CHECK("bce_multi_forward");
  return 0;
}
extern "C" int s2i_bce_multi_backward(const float* const* probs, const float* target, const float* weight, int G, int H,
                                      int B, const float* gout, float* const* dprobs, void* stream) {
  S2I_REQUIRE(probs && dprobs && target && weight && gout && G > 0 && H > 0 && H <= 4 && B > 0,
              "bce_multi_backward: bad args");
  MultiPtr mp = {};
  for (int h = 0; h < H; ++h) {
    S2I_REQUIRE(probs[h] && dprobs[h], "bce_multi_backward: null head");
    mp.p[h] = probs[h];
    mp.d[h] = dprobs[h];
  }
  hipLaunchKernelGGL(bce_multi_bwd_kernel, dim3((G * H * B + 255) / 256), dim3(256), 0, ST, mp, target, weight, G, H, B,
                     gout);
  S2I_LAUNCH_CHECK("bce_multi_backward");
  return 0;
}

extern "C" int s2i_cal_loss(const float* scores, const int* labels, int B, int D, float* loss, int accumulate,
                            float* dscores_sym, void* stream) {
  S2I_REQUIRE(scores && labels && loss && B > 0 && D > 0, "cal_loss: bad args");
  hipLaunchKernelGGL(cal_loss_kernel, dim3(1), dim3(256), 0, ST, scores, labels, B, D, loss, accumulate, dscores_sym);
  S2I_LAUNCH_CHECK("cal_loss");
  return 0;
}

extern "C" int s2i_maxpool_w3s2(const float* x, int B, int H, int W, int C, float* y, void* stream) {
  S2I_REQUIRE(x && y && B > 0 && H > 0 && W >= 2 && W % 2 == 0 && C > 0 && C % 4 == 0, "maxpool_w3s2: bad args");
  const long long total = (long long)B * H * (W / 2) * (C / 4);
  hipLaunchKernelGGL(maxpool_w3s2_kernel, dim3(grid_for(total)), dim3(256), 0, ST, x, W, C, total, y);
  S2I_LAUNCH_CHECK("maxpool_w3s2");
  return 0;
}
extern "C" int s2i_lstm_cell(const float* xproj, int ldx, const float* hproj, const int* lens, int B, int T, int Hd,
                             int step, int reverse, float* h, float* c, float* out, int ldo, void* stream) {
  S2I_REQUIRE(xproj && hproj && lens && h && c && out && B > 0 && T > 0 && Hd > 0 && step >= 0 && step < T,
              "lstm_cell: bad args");
  S2I_REQUIRE(ldx >= 4 * Hd && ldo >= Hd, "lstm_cell: row strides too small");
  hipLaunchKernelGGL(lstm_cell_kernel, dim3((B * Hd + 255) / 256), dim3(256), 0, ST, xproj, ldx, hproj, lens, B, T, Hd,
                     step, reverse, h, c, out, ldo);
  S2I_LAUNCH_CHECK("lstm_cell");
  return 0;
}
extern "C" int s2i_lstm_step(const float* xproj, int ldx, const float* whh_fwd, const float* whh_rev, const int* lens,
                             int B, int T, int Hd, int D, int step, const float* h_in, float* h_out, float* c, float* out,
                             int ldo, void* stream) {
  S2I_REQUIRE(xproj && whh_fwd && lens && h_in && h_out && c && out && h_in != h_out, "lstm_step: bad pointers");
  S2I_REQUIRE((D == 1 || (D == 2 && whh_rev)) && B > 0 && B <= 32 && T > 0 && step >= 0 && Hd > 0 && (Hd % 8) == 0 &&
                  Hd <= 512, "lstm_step: unsupported extents (B=%d Hd=%d D=%d)", B, Hd, D);
  S2I_REQUIRE(ldx >= D * 4 * Hd && ldo >= D * Hd, "lstm_step: row strides too small");
  const size_t shb = (size_t)2 * 32 * (Hd + 4) * sizeof(float);
  S2I_REQUIRE(shb <= 160 * 1024, "lstm_step: Hd=%d needs %zu bytes of LDS", Hd, shb);
  static bool attr_set = false;
  if (shb > 65536 && !attr_set) {
    S2I_REQUIRE(hipFuncSetAttribute((const void*)lstm_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) ==
                    hipSuccess, "lstm_step: cannot raise the dynamic LDS limit");
    attr_set = true;
  }
  hipLaunchKernelGGL(lstm_step_kernel, dim3(Hd / 8, D), dim3(256), shb, ST, xproj, ldx, whh_fwd, whh_rev, lens, B, T, Hd,
                     step, h_in, h_out, c, out, ldo);
  S2I_LAUNCH_CHECK("lstm_step");
  return 0;
}
extern "C" int s2i_time_mean(const float* x, int B, int T, int C, float* y, void* stream) {
  S2I_REQUIRE(x && y && B > 0 && T > 0 && C > 0, "time_mean: bad args");
  hipLaunchKernelGGL(time_mean_kernel, dim3((B * C + 255) / 256), dim3(256), 0, ST, x, B, T, C, y);
  S2I_LAUNCH_CHECK("time_mean");
  return 0;
}

extern "C" int s2i_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                             float beta2, float eps, int step, const int* step_dev, float gscale, void* stream) {
  S2I_REQUIRE(p && g && m && v && n > 0, "adam_step: bad args");
  S2I_REQUIRE(step_dev || step >= 1, "adam_step: step must be >= 1");
  const long long n4 = (n + 3) / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4)), dim3(256), 0, ST, p, g, m, v, n4, n, lr, beta1, beta2, eps,
                     step, step_dev, gscale);
  S2I_LAUNCH_CHECK("adam_step");
  return 0;
}
extern "C" int s2i_increment(int* counter, void* stream) {
  S2I_REQUIRE(counter, "increment: null");
  hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(1), 0, ST, counter);
  S2I_LAUNCH_CHECK("increment");
  return 0;
}
extern "C" int s2i_ema_update(float* avg, const float* p, long long n, float decay, void* stream) {
  S2I_REQUIRE(avg && p && n > 0, "ema_update: bad args");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, ST, avg, p, n, decay);
  S2I_LAUNCH_CHECK("ema_update");
  return 0;
}
extern "C" int s2i_scale_dev(float* y, const float* x, long long n, const float* a_dev, void* stream) {
  S2I_REQUIRE(y && x && a_dev && n > 0, "scale_dev: bad args");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, ST, y, x, n, a_dev);
  S2I_LAUNCH_CHECK("scale_dev");
  return 0;
}
extern "C" int s2i_axpby(float* y, const float* x, long long n, float a, float b, void* stream) {
  S2I_REQUIRE(y && x && n > 0, "axpby: bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, ST, y, x, n, a, b);
  S2I_LAUNCH_CHECK("axpby");
  return 0;
}
