// Implicit-GEMM convolution kernels for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_32x32x2_f32, bit-exact f32 fma chain).
//
// One gather formulation covers every convolution on the StackGAN-v2 path
// (reference StackGAN_v2/model.py:125-140, 144-169, 287-298, 358-398):
//   K1    : 1x1 / nn.Linear
//   K3S1  : conv3x3 pad 1 (and, with flipped taps + transposed weights, its input gradient)
//   K4S2  : Conv2d(k4,s2,p1) of the D towers; also the input gradient of an upBlock
//   TCONV : 4-phase transposed k4 s2 p1 conv = nearest-x2 upsample + conv3x3 collapsed to 2x2 taps
//           per output parity (2.25x fewer MACs than the literal upsample+conv); also the input
//           gradient of Conv2d(k4,s2,p1)
// Activations are NHWC so the K (channel) direction of the gather is contiguous in HBM; a per-image
// vector (c_code) can be concatenated in front of the stored channels without materialising the
// torch.cat of model.py:277/434.
//
// Tile: 256 threads = 4 waves; block tile BM x BN x 32; each wave owns TM x TN MFMA tiles of 32x32.
// LDS holds A as [k][m] and B as [k][n] so a fragment read is 32 consecutive floats per half-wave
// (ds_read_b32, conflict-free).  Global->LDS staging goes through registers with the next chunk's
// loads in flight during the MFMA loop.
#include "s2i_common.h"

namespace {

struct IgemmP {
  const float* __restrict__ x;
  const float* __restrict__ cvec;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  const float* __restrict__ cls_bias;
  float* __restrict__ y;
  float* __restrict__ part;
  float* __restrict__ slab;
  int B, H, W, Cx, Cc, Ca;
  int Ho, Wo, lgWo, lgHoWo;
  int M, N, K, T;
  int kind, flip, act, stats, splitk, cps, nchunks;
  int ldw, wR, ldy, nparts;
  int g_kw, g_s, g_pad;  // geometry of S2I_CONV_1D (1 x kw taps along W, stride, padding)
  int wt;                // weights read transposed per tap (small_n_conv_kernel; the igemm takes it as a template flag)
  int x16, y16;          // x / y hold bf16 instead of fp32 (bf16 activation mode: the arithmetic here stays fp32)
  unsigned x_bytes, c_bytes, w_bytes;
  long long Mrows;
  // split-bf16 weights [plane][tap][n][k] (igemm_fwd_split_kernel)
  const unsigned short* __restrict__ wsp;
  int wsp_np, wsp_kp, wsp_plane;  // rows per tap, row length (bf16 elements), elements per plane
  unsigned wsp_bytes;
  // apply-on-load (INACT instantiations): x holds the RAW output of the producing convolution; the gather applies that
  // layer's BatchNorm (scale, shift from its (groups, 4, Cx) coefficient table) and LeakyReLU while it stages the operand
  const float* __restrict__ in_coef;
  int in_rows_per_group;   // output rows of THIS launch per BatchNorm group of the producer (rows beyond: next group)
};

__device__ __forceinline__ void geom(const IgemmP& p, int kind, int& s, int& pad, int& kw) {
  if (kind == S2I_CONV_1D) { s = p.g_s; pad = p.g_pad; kw = p.g_kw; }
  else if (kind == S2I_CONV_K3S1) { s = 1; pad = 1; kw = 3; }
  else if (kind == S2I_CONV_K4S2) { s = 2; pad = 1; kw = 4; }
  else { s = 1; pad = 0; kw = 1; }
}

__device__ __forceinline__ void tap_delta(int kind, int kw, int t, int py, int px, int& dy, int& dx) {
  if (kind == S2I_TCONV_K4S2) {
    const int a = t >> 1, b = t & 1;
    dy = a ? (py ? 1 : -1) : 0;
    dx = b ? (px ? 1 : -1) : 0;
  } else if (kind == S2I_CONV_1D) {
    dy = 0;
    dx = t;
  } else {
    dy = t / kw;
    dx = t - dy * kw;
  }
}

__device__ __forceinline__ void geom(int kind, int& s, int& pad, int& kw) {
  if (kind == S2I_CONV_K3S1) { s = 1; pad = 1; kw = 3; }
  else if (kind == S2I_CONV_K4S2) { s = 2; pad = 1; kw = 4; }
  else { s = 1; pad = 0; kw = 1; }
}

// bit t set <=> tap t of the pixel whose base coordinate is (by,bx) falls inside the H x W tensor
__device__ __forceinline__ unsigned tap_mask(int kind, int kw, int by, int bx, int H, int W, int py, int px) {
  if (kind == S2I_TCONV_K4S2) {
    const int sy = py ? 1 : -1, sx = px ? 1 : -1;
    const bool y0 = by >= 0 && by < H, y1 = by + sy >= 0 && by + sy < H;
    const bool x0 = bx >= 0 && bx < W, x1 = bx + sx >= 0 && bx + sx < W;
    return (unsigned)(y0 && x0) | ((unsigned)(y0 && x1) << 1) | ((unsigned)(y1 && x0) << 2) |
           ((unsigned)(y1 && x1) << 3);
  }
  unsigned cols = 0, mask = 0;
  for (int kx = 0; kx < kw; ++kx) cols |= (unsigned)(bx + kx >= 0 && bx + kx < W) << kx;
  if (kind == S2I_CONV_1D) return (by >= 0 && by < H) ? cols : 0u;
  for (int ky = 0; ky < kw; ++ky)
    if (by + ky >= 0 && by + ky < H) mask |= cols << (ky * kw);
  return mask;
}

// which tap of the packed weight tensor the gather tap t multiplies
__device__ __forceinline__ int tap_weight(int kind, int flip, int T, int t, int py, int px) {
  if (kind == S2I_TCONV_K4S2) {
    const int a = t >> 1, b = t & 1;
    const int k4y = py ? (a ? 0 : 2) : (a ? 3 : 1);
    const int k4x = px ? (b ? 0 : 2) : (b ? 3 : 1);
    return k4y * 4 + k4x;
  }
  return flip ? (T - 1 - t) : t;
}

// One 32-deep K chunk: 16 k-pairs, each TM x TN v_mfma_f32_32x32x2_f32.  The fragments of pair kk+1 are read
// from LDS BEFORE the MFMAs of pair kk are issued (two register sets), so the LDS latency sits behind 4+ MFMAs
// of this wave instead of relying on the other waves of the SIMD to cover it.
template <int TM, int TN, int LDA, int LDB, int KK0 = 0, int KK1 = 16>
__device__ __forceinline__ void mma_chunk(const float* As, const float* Bs, int arow0, int bcol0,
                                          int lane, f32x16 (&acc)[TM][TN]) {
  static_assert((KK1 - KK0) % 2 == 0, "k-pairs are processed two at a time");
  const int l31 = lane & 31, lh = lane >> 5;
  const float* ap = As + lh * LDA + arow0 + l31;
  const float* bp = Bs + lh * LDB + bcol0 + l31;
  float a0[TM], b0[TN], a1[TM], b1[TN];  // two named fragment sets (a runtime-indexed pair would go to scratch)
#pragma unroll
  for (int i = 0; i < TM; ++i) a0[i] = ap[(2 * KK0) * LDA + i * 32];
#pragma unroll
  for (int j = 0; j < TN; ++j) b0[j] = bp[(2 * KK0) * LDB + j * 32];
#pragma unroll
  for (int kk = KK0; kk < KK1; kk += 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i) a1[i] = ap[(2 * (kk + 1)) * LDA + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) b1[j] = bp[(2 * (kk + 1)) * LDB + j * 32];
    // pin the order (hipcc otherwise sinks the reads next to their use): next pair's LDS reads, THEN this pair's
    // MFMAs.  Only for the 2x2 wave tile: with fewer MFMAs per pair the pinned schedule makes hipcc spill.
    if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
    if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
    if (kk + 2 < KK1) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a0[i] = ap[(2 * (kk + 2)) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b0[j] = bp[(2 * (kk + 2)) * LDB + j * 32];
    }
    if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
    if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define S2I_OOB 0x7ffffff0  // byte offset past any tensor: the buffer bounds check returns zeros

__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
// four consecutive elements at the byte offset an fp32 tensor would have; `is16`: the tensor holds bf16 (half the offset)
__device__ __forceinline__ f32x4 bload4_any(__amdgpu_buffer_rsrc_t r, int byte_off, int is16) {
  if (!is16) return bload4(r, byte_off);
  const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off == S2I_OOB ? S2I_OOB : (byte_off >> 1), 0, 0);
  return f32x4{__builtin_bit_cast(float, v[0] << 16), __builtin_bit_cast(float, v[0] & 0xffff0000u),
               __builtin_bit_cast(float, v[1] << 16), __builtin_bit_cast(float, v[1] & 0xffff0000u)};
}
__device__ __forceinline__ unsigned short f2bf(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

// CA32: gathered channel count (and the broadcast-vector part of it) is a multiple of 32, so a 32-deep K
// chunk lies inside ONE tap (and entirely in x or entirely in cvec): the tap decode is scalar work.
// Measured alternatives that lost (MI355X, 64->128 k4s2 on 24x128x128): two LDS stages with one barrier per
// chunk (2 blocks/CU: 79 vs 98 TFLOP/s), a start-up stagger of the blocks (no change).  Three resident blocks
// per CU with the plain two-barrier loop is the fastest structure found for v_mfma_f32_32x32x2_f32.
// INACT (CA32, no broadcast vector, forward weights only): see IgemmP::in_coef.  Padding taps stay zero AFTER the activation.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool WT, bool CA32, bool INACT = false>
__global__ __launch_bounds__(256, 3) void igemm_fwd_kernel(IgemmP p) {
  static_assert(!INACT || (CA32 && !WT), "apply-on-load: 32-channel chunks, forward weight layout");
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int LDA = BM + 1;
  constexpr int LDB = WT ? BN + 1 : BN;
  constexpr int ASLOTS = BM / 32;
  constexpr int BSLOTS = BN / 32;
  constexpr int BROWS_PER_PASS = 1024 / BN;
  // 96-row tiles are padded to the LDS footprint of a 128-row tile: exactly three blocks per CU either way, so that
  // 768 blocks are one round of the chip for both (plan_fwd counts rounds)
  constexpr int SMEM_FLOATS = 32 * LDA + 32 * LDB + 4;
  constexpr int SMEM_MIN = BM < 128 ? 32 * 129 + 32 * LDB + 4 : 0;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS > SMEM_MIN ? SMEM_FLOATS : SMEM_MIN];
  float* As = smem + (WT ? 0 : 32 * LDB);  // keep the b128-written array 16-byte aligned
  float* Bs = smem + (WT ? 32 * LDA : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  // XCD-aware block order (round 3): the column blocks, phases and K splits of ONE row tile gather the same input pixels;
  // launch order puts them gridDim.x ids apart (another XCD's L2, another time).  Linear id -> (row tile, sibling) with all
  // siblings of a row tile on one XCD (same id % 8) and adjacent there; a bijection for any grid.
  int bx, by, bz;
  {
    const int sib = gridDim.y * gridDim.z;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int per_group = 8 * sib;
    const int grp = L / per_group, Ll = L - grp * per_group;
    const int in_group = min(8, (int)gridDim.x - grp * 8);
    bx = grp * 8 + Ll % in_group;
    const int u = Ll / in_group;
    by = u % gridDim.y;
    bz = u / gridDim.y;
  }
  int phase = 0, split = bz;
  if (p.kind == S2I_TCONV_K4S2) { phase = bz / p.splitk; split = bz - phase * p.splitk; }
  const int py = phase >> 1, px = phase & 1;
  const int m0 = bx * BM, n0 = by * BN;
  const int kq = tid & 7, mrow = tid >> 3;
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);

  // hardware-bounds-checked descriptors: an invalid element is fetched at S2I_OOB and reads as zero,
  // so the gather needs no exec-mask branches
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.cvec, 0, p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  int aoff[ASLOTS];     // byte offset of the slot's base pixel (may be negative; only used when in bounds)
  unsigned amask[ASLOTS];
  int acoff[ASLOTS];    // byte offset of the slot's row of cvec
#pragma unroll
  for (int i = 0; i < ASLOTS; ++i) {
    const int m = m0 + mrow + 32 * i;
    unsigned mask = 0;
    int base = 0, coff = 0;
    if (m < p.M) {
      const int b = m >> p.lgHoWo;
      const int r = m & ((1 << p.lgHoWo) - 1);
      const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
      const int by = p.kind == S2I_CONV_1D ? oy : oy * s - pad, bx = ox * s - pad;
      mask = tap_mask(p.kind, kw, by, bx, p.H, p.W, py, px);
      base = (((b * p.H + by) * p.W + bx) * p.Cx + kq * 4) * 4;
      coff = (b * p.Cc + kq * 4) * 4;
    }
    aoff[i] = base;
    amask[i] = mask;
    acoff[i] = coff;
  }
  // per-thread constant parts of the weight addresses
  const int bcol4 = tid % (BN / 4), brow = tid / (BN / 4);
  int wconst[BSLOTS];
#pragma unroll
  for (int j = 0; j < BSLOTS; ++j) {
    if (WT) {
      const int n = n0 + mrow + 32 * j;
      wconst[j] = n < p.N ? (n * p.ldw + kq * 4) * 4 : S2I_OOB;
    } else {
      const int n = n0 + bcol4 * 4;
      wconst[j] = n < p.ldw ? ((brow + j * BROWS_PER_PASS) * p.ldw + n) * 4 : S2I_OOB;
    }
  }

  f32x4 ra[ASLOTS], rb[BSLOTS];
  // apply-on-load state of the chunk held in ra: scale / shift of this thread's four channels and the chunk's tap
  f32x4 in_s = {1.f, 1.f, 1.f, 1.f}, in_t = {0.f, 0.f, 0.f, 0.f};
  int in_tap = 0;
  const float* in_cg = nullptr;
  if constexpr (INACT) in_cg = p.in_coef + (size_t)(m0 / p.in_rows_per_group) * 4 * p.Cx;   // a tile lies inside one group

  auto fetch = [&](int kc) {
    if (CA32) {
      // wave-uniform tap decode
      const int k0 = kc * 32;
      const int t = k0 / p.Ca;
      const int c0 = k0 - t * p.Ca;
      int dy, dx;
      tap_delta(p.kind, kw, t, py, px, dy, dx);
      const int tw = tap_weight(p.kind, p.flip, p.T, t, py, px);
      if constexpr (INACT) {
        in_tap = t;
        in_s = *reinterpret_cast<const f32x4*>(in_cg + 2 * p.Cx + c0 + kq * 4);
        in_t = *reinterpret_cast<const f32x4*>(in_cg + 3 * p.Cx + c0 + kq * 4);
      }
      if (c0 < p.Cc) {
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i)
          ra[i] = bload4(rc, ((amask[i] >> t) & 1u) ? acoff[i] + c0 * 4 : S2I_OOB);
      } else {
        const int toff = ((dy * p.W + dx) * p.Cx + (c0 - p.Cc)) * 4;
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i)
          ra[i] = bload4_any(rx, ((amask[i] >> t) & 1u) ? aoff[i] + toff : S2I_OOB, p.x16);
      }
      const int wbase = WT ? (tw * p.wR * p.ldw + c0) * 4 : (tw * p.wR + c0) * p.ldw * 4;
#pragma unroll
      for (int j = 0; j < BSLOTS; ++j) rb[j] = bload4(rw, wconst[j] == S2I_OOB ? S2I_OOB : wbase + wconst[j]);
    } else {
      const int k = kc * 32 + kq * 4;
      const bool kvalid = k < p.K;
      int t = 0, c = 0, tw = 0, toff = 0;
      if (kvalid) {
        t = k / p.Ca;
        c = k - t * p.Ca;
        int dy, dx;
        tap_delta(p.kind, kw, t, py, px, dy, dx);
        toff = ((dy * p.W + dx) * p.Cx + (c - kq * 4 - p.Cc)) * 4;
        tw = tap_weight(p.kind, p.flip, p.T, t, py, px);
      }
      const bool from_vec = c < p.Cc;
#pragma unroll
      for (int i = 0; i < ASLOTS; ++i) {
        const bool ok = kvalid && ((amask[i] >> t) & 1u);
        f32x4 vx = bload4_any(rx, (ok && !from_vec) ? aoff[i] + toff : S2I_OOB, p.x16);
        if (p.Cc > 0) vx += bload4(rc, (ok && from_vec) ? acoff[i] + (c - kq * 4) * 4 : S2I_OOB);
        ra[i] = vx;
      }
      if (WT) {
#pragma unroll
        for (int j = 0; j < BSLOTS; ++j)
          rb[j] = bload4(rw, (kvalid && wconst[j] != S2I_OOB) ? (tw * p.wR * p.ldw + c - kq * 4) * 4 + wconst[j] : S2I_OOB);
      } else {
#pragma unroll
        for (int q = 0; q < BSLOTS; ++q) {
          const int kb = kc * 32 + brow + q * BROWS_PER_PASS;
          int off = S2I_OOB;
          if (kb < p.K && wconst[q] != S2I_OOB) {
            const int tb = kb / p.Ca;
            const int cb = kb - tb * p.Ca;
            const int twb = tap_weight(p.kind, p.flip, p.T, tb, py, px);
            off = ((twb * p.wR + cb) * p.ldw + n0 + bcol4 * 4) * 4;
          }
          rb[q] = bload4(rw, off);
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunks, c_begin + p.cps);
  auto stage_store = [&](float* Asd, float* Bsd) {
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = ra[i][j];
        if constexpr (INACT) {
          v = fmaf(v, in_s[j], in_t[j]);
          v = v > 0.f ? v : 0.2f * v;
          v = ((amask[i] >> in_tap) & 1u) ? v : 0.f;      // the padding is zero in the ACTIVATED tensor
        }
        Asd[(kq * 4 + j) * LDA + mrow + 32 * i] = v;
      }
    if (WT) {
#pragma unroll
      for (int i = 0; i < BSLOTS; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bsd[(kq * 4 + j) * LDB + mrow + 32 * i] = rb[i][j];
    } else {
#pragma unroll
      for (int q = 0; q < BSLOTS; ++q)
        *reinterpret_cast<f32x4*>(Bsd + (brow + q * BROWS_PER_PASS) * LDB + bcol4 * 4) = rb[q];
    }
  };
  {
    if (c_begin < c_end) fetch(c_begin);
    for (int kc = c_begin; kc < c_end; ++kc) {
      stage_store(As, Bs);
      __syncthreads();
      if (kc + 1 < c_end) fetch(kc + 1);
      mma_chunk<TM, TN, LDA, LDB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
      __syncthreads();
    }
  }

  // ---- epilogue ----
  const int l31 = lane & 31, lh = lane >> 5;
  const bool tconv = p.kind == S2I_TCONV_K4S2;
  const bool raw = p.splitk > 1;
  if (p.cls_bias && !raw) {
    // contribution of a spatially constant operand (the broadcast c_code of model.py:277), pre-reduced per
    // border class: cls = 3 * (top | middle | bottom) + (left | middle | right)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M) continue;
        const int b = m >> p.lgHoWo;
        const int rr = m & ((1 << p.lgHoWo) - 1);
        const int oy = rr >> p.lgWo, ox = rr & (p.Wo - 1);
        const int cls = 3 * (oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1)) + (ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1));
        const float* bp = p.cls_bias + ((size_t)b * 9 + cls) * p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * TN * 32 + j * 32 + l31;
          if (n < p.N) acc[i][j][r] += bp[n];
        }
      }
  }
  float* outp = raw ? p.slab + (size_t)split * p.Mrows * p.N : p.y;
  const int ldo = raw ? p.N : p.ldy;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int m = m0 + ml;
      if (m >= p.M) continue;
      long long row = m;
      if (tconv) {
        const int b = m >> p.lgHoWo;
        const int rr = m & ((1 << p.lgHoWo) - 1);
        const int oy = rr >> p.lgWo, ox = rr & (p.Wo - 1);
        row = ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n < p.N) {
          float v = acc[i][j][r];
          if (!raw) {
            if (p.bias) v += p.bias[n];
            if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
            else if (p.act == S2I_ACT_TANH) v = tanhf(v);
            else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
          }
          if (!raw && p.y16) reinterpret_cast<unsigned short*>(outp)[row * ldo + n] = f2bf(v);
          else outp[row * ldo + n] = v;
        }
      }
    }
  }

  if (p.stats && !raw) {
    // column sums over this block's rows; rows >= M gathered zeros and contribute nothing
    float* red = smem;  // [2][WAVES_M][BN]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sv = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          sv += v;
          sq += v * v;
        }
      sv += __shfl_xor(sv, 32);
      sq += __shfl_xor(sq, 32);
      if (lh == 0) {
        const int col = wn * TN * 32 + j * 32 + l31;
        red[(0 * WAVES_M + wm) * BN + col] = sv;
        red[(1 * WAVES_M + wm) * BN + col] = sq;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      if (n < p.N) {
        float sv = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < WAVES_M; ++q) {
          sv += red[(0 * WAVES_M + q) * BN + tid];
          sq += red[(1 * WAVES_M + q) * BN + tid];
        }
        const int gm = phase * gridDim.x + bx;
        p.part[((size_t)0 * p.nparts + gm) * p.N + n] = sv;
        p.part[((size_t)1 * p.nparts + gm) * p.N + n] = sq;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Split-bf16 variant ("bf16xNP" math modes): every fp32 operand value v is written as the sum of NP bf16 numbers
// (v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2)) and the products a_i * b_j with i + j <= NP + 1 are
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, which runs at 16x the rate of v_mfma_f32_32x32x2_f32.
//   NP = 2: 3 products, relative product error ~2^-16 (TF32, which the reference's cuDNN convolutions use by default on
//           NVIDIA hardware, is 2^-11);   NP = 3: 6 products, ~2^-23.
// Activations are split while they are staged into LDS (v_cvt_pk_bf16_f32 + two VALU ops per extra plane and pair);
// weights arrive pre-split in [plane][tap][n][k] order (s2i_split_packed_weight), so a B tile is a straight 16-byte copy.
// LDS: one [rows][32 k] bf16 image per plane and operand, 64-byte rows whose four 16-byte segments are XOR-swizzled
// with (row >> 2) & 3: a fragment is ONE ds_read_b128 per lane (row r = lane & 31, k = 8 * (lane >> 5) + j) and 16
// consecutive rows hit 16 distinct 4-bank groups; three planes of a 128x128 tile take 48 KB, so three blocks fit a CU.
// Gather, split-K, epilogue and statistics are those of igemm_fwd_kernel (CA32 case only).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int NP>
__device__ __forceinline__ void split4(const f32x4 v, u32x2 (&out)[NP]) {
  f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    const unsigned pa = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2));
    const unsigned pb = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2));
    out[pl] = u32x2{pa, pb};
    if (pl + 1 < NP) {
      a[0] -= __builtin_bit_cast(float, pa << 16);
      a[1] -= __builtin_bit_cast(float, pa & 0xffff0000u);
      b[0] -= __builtin_bit_cast(float, pb << 16);
      b[1] -= __builtin_bit_cast(float, pb & 0xffff0000u);
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NP>
__global__ __launch_bounds__(256, 3) void igemm_fwd_split_kernel(IgemmP p) {
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int ROWB = 64;
  constexpr int APLANE = BM * ROWB, BPLANE = BN * ROWB;
  constexpr int ASLOTS = BM / 32;
  constexpr int BSEGS = NP * BN * 4;                 // 16-byte segments of the B tile
  constexpr int BLOADS = (BSEGS + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NP * (APLANE + BPLANE)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + NP * APLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int phase = 0, split = blockIdx.z;
  if (p.kind == S2I_TCONV_K4S2) { phase = blockIdx.z / p.splitk; split = blockIdx.z - phase * p.splitk; }
  const int py = phase >> 1, px = phase & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kq = tid & 7, mrow = tid >> 3;
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)p.cvec, 0, p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wsp, 0, p.wsp_bytes, 0x00020000);

  int aoff[ASLOTS];
  unsigned amask[ASLOTS];
  int acoff[ASLOTS];
#pragma unroll
  for (int i = 0; i < ASLOTS; ++i) {
    const int m = m0 + mrow + 32 * i;
    unsigned mask = 0;
    int base = 0, coff = 0;
    if (m < p.M) {
      const int b = m >> p.lgHoWo;
      const int r = m & ((1 << p.lgHoWo) - 1);
      const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
      const int by = p.kind == S2I_CONV_1D ? oy : oy * s - pad, bx = ox * s - pad;
      mask = tap_mask(p.kind, kw, by, bx, p.H, p.W, py, px);
      base = (((b * p.H + by) * p.W + bx) * p.Cx + kq * 4) * 4;
      coff = (b * p.Cc + kq * 4) * 4;
    }
    aoff[i] = base;
    amask[i] = mask;
    acoff[i] = coff;
  }
  int bconst[BLOADS], blds[BLOADS];
#pragma unroll
  for (int q = 0; q < BLOADS; ++q) {
    const int e = tid + q * 256;
    const int seg = e & 3, row = (e >> 2) % BN, pl = (e >> 2) / BN;
    const int n = n0 + row;
    bconst[q] = (e < BSEGS && n < p.N) ? (pl * p.wsp_plane + n * p.wsp_kp) * 2 + seg * 16 : S2I_OOB;
    blds[q] = e < BSEGS ? pl * BPLANE + row * ROWB + ((seg ^ ((row >> 2) & 3)) << 4) : -1;
  }

  f32x4 ra[ASLOTS];
  u32x4 rb[BLOADS];
  auto fetch = [&](int kc) {
    const int k0 = kc * 32;
    const int t = k0 / p.Ca;
    const int c0 = k0 - t * p.Ca;
    int dy, dx;
    tap_delta(p.kind, kw, t, py, px, dy, dx);
    const int tw = tap_weight(p.kind, p.flip, p.T, t, py, px);
    if (c0 < p.Cc) {
#pragma unroll
      for (int i = 0; i < ASLOTS; ++i)
        ra[i] = bload4(rc, ((amask[i] >> t) & 1u) ? acoff[i] + c0 * 4 : S2I_OOB);
    } else {
      const int toff = ((dy * p.W + dx) * p.Cx + (c0 - p.Cc)) * 4;
#pragma unroll
      for (int i = 0; i < ASLOTS; ++i)
        ra[i] = bload4(rx, ((amask[i] >> t) & 1u) ? aoff[i] + toff : S2I_OOB);
    }
    const int wbase = (tw * p.wsp_np * p.wsp_kp + c0) * 2;
#pragma unroll
    for (int q = 0; q < BLOADS; ++q)
      rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rw, bconst[q] == S2I_OOB ? S2I_OOB : wbase + bconst[q], 0, 0);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  const int fsw = ((lh ^ ((l31 >> 2) & 3)) << 4);  // swizzled segment of k-step 0; k-step 1 is fsw ^ 32
  const unsigned char* ap = As + (wm * TM * 32 + l31) * ROWB;
  const unsigned char* bp = Bs + (wn * TN * 32 + l31) * ROWB;

  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunks, c_begin + p.cps);
  if (c_begin < c_end) fetch(c_begin);
  for (int kc = c_begin; kc < c_end; ++kc) {
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
      u32x2 sp[NP];
      split4<NP>(ra[i], sp);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        *reinterpret_cast<u32x2*>(As + pl * APLANE + (mrow + 32 * i) * ROWB + ((((kq >> 1) ^ (mrow >> 2)) & 3) << 4) +
                                  (kq & 1) * 8) = sp[pl];
    }
#pragma unroll
    for (int q = 0; q < BLOADS; ++q)
      if (blds[q] >= 0) *reinterpret_cast<u32x4*>(Bs + blds[q]) = rb[q];
    // the MFMA phase of a split chunk is short (0.7 us): put the next chunk's loads in flight before the barrier wait
    if (kc + 1 < c_end) fetch(kc + 1);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = fsw ^ (ks << 5);
      bf16x8 a[NP][TM];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[pl][i] = *reinterpret_cast<const bf16x8*>(ap + pl * APLANE + i * 32 * ROWB + so);
      // b plane by plane, the smallest cross terms first: (a1 b3) | (a2 b2, a1 b2) | (a3 b1, a2 b1, a1 b1)
#pragma unroll
      for (int pb = NP - 1; pb >= 0; --pb) {
        bf16x8 b[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(bp + pb * BPLANE + j * 32 * ROWB + so);
#pragma unroll
        for (int pa = NP - 1 - pb; pa >= 0; --pa)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue (as igemm_fwd_kernel) ----
  const bool tconv = p.kind == S2I_TCONV_K4S2;
  const bool raw = p.splitk > 1;
  if (p.cls_bias && !raw) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.M) continue;
        const int b = m >> p.lgHoWo;
        const int rr = m & ((1 << p.lgHoWo) - 1);
        const int oy = rr >> p.lgWo, ox = rr & (p.Wo - 1);
        const int cls = 3 * (oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1)) + (ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1));
        const float* bpt = p.cls_bias + ((size_t)b * 9 + cls) * p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * TN * 32 + j * 32 + l31;
          if (n < p.N) acc[i][j][r] += bpt[n];
        }
      }
  }
  float* outp = raw ? p.slab + (size_t)split * p.Mrows * p.N : p.y;
  const int ldo = raw ? p.N : p.ldy;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= p.M) continue;
      long long row = m;
      if (tconv) {
        const int b = m >> p.lgHoWo;
        const int rr = m & ((1 << p.lgHoWo) - 1);
        const int oy = rr >> p.lgWo, ox = rr & (p.Wo - 1);
        row = ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n < p.N) {
          float v = acc[i][j][r];
          if (!raw) {
            if (p.bias) v += p.bias[n];
            if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
            else if (p.act == S2I_ACT_TANH) v = tanhf(v);
            else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
          }
          outp[row * ldo + n] = v;
        }
      }
    }
  }
  if (p.stats && !raw) {
    float* red = reinterpret_cast<float*>(smem);  // [2][WAVES_M][BN]
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sv = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          sv += v;
          sq += v * v;
        }
      sv += __shfl_xor(sv, 32);
      sq += __shfl_xor(sq, 32);
      if (lh == 0) {
        const int col = wn * TN * 32 + j * 32 + l31;
        red[(0 * WAVES_M + wm) * BN + col] = sv;
        red[(1 * WAVES_M + wm) * BN + col] = sq;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      if (n < p.N) {
        float sv = 0.f, sq = 0.f;
#pragma unroll
        for (int q = 0; q < WAVES_M; ++q) {
          sv += red[(0 * WAVES_M + q) * BN + tid];
          sq += red[(1 * WAVES_M + q) * BN + tid];
        }
        const int gm = phase * gridDim.x + blockIdx.x;
        p.part[((size_t)0 * p.nparts + gm) * p.N + n] = sv;
        p.part[((size_t)1 * p.nparts + gm) * p.N + n] = sq;
      }
    }
  }
}

// fp32 packed weights P[T][R][C] -> NP bf16 planes in BOTH operand layouts from one read:
//   dst_rc [plane][T][R][C] (input gradient: n = r, k = c) and dst_cr [plane][T][C][R] (forward: n = c, k = r); either may be null
__global__ __launch_bounds__(256) void split_packed_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst_rc,
                                                           unsigned short* __restrict__ dst_cr, int R, int C, int NP,
                                                           long long plane) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* sp = src + (size_t)t * R * C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? sp[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (dst_rc) {
      const int r = r0 + ty + 8 * k, c = c0 + tx;
      if (r < R && c < C) {
        float v = tile[ty + 8 * k][tx];
        const size_t o = ((size_t)t * R + r) * C + c;
        for (int pl = 0; pl < NP; ++pl) {
          const __bf16 h = (__bf16)v;
          dst_rc[pl * plane + o] = __builtin_bit_cast(unsigned short, h);
          v -= (float)h;
        }
      }
    }
    if (dst_cr) {
      const int c = c0 + ty + 8 * k, r = r0 + tx;
      if (r < R && c < C) {
        float v = tile[tx][ty + 8 * k];
        const size_t o = ((size_t)t * C + c) * R + r;
        for (int pl = 0; pl < NP; ++pl) {
          const __bf16 h = (__bf16)v;
          dst_cr[pl * plane + o] = __builtin_bit_cast(unsigned short, h);
          v -= (float)h;
        }
      }
    }
  }
}

// Convolutions with at most 4 output channels (GET_IMAGE_G's conv3x3 -> RGB, model.py:287-298, and the input
// gradient of the discriminators' first conv): HBM-bound, so no matrix cores.  LPP = Ca/4 lanes share one output
// pixel, each multiplying its 4 input channels into the 4 outputs (weights [t][c][4] in LDS), then a shuffle
// reduction; a wave reads PPW = 64/LPP whole pixels per tap, i.e. contiguous NHWC bytes.
template <int LPP>
__global__ __launch_bounds__(256) void small_n_conv_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [T][Ca][4]
  constexpr int PPW = 64 / LPP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int phase = blockIdx.z;
  const int py = phase >> 1, px = phase & 1;
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);
  // stage the 4 output columns of every (tap, channel) row
  for (int e = tid; e < p.T * p.Ca; e += 256) {
    const int t = e / p.Ca, c = e - t * p.Ca;
    const int tw = tap_weight(p.kind, p.flip, p.T, t, py, px);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p.wt) {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (n < p.N) v[n] = p.w[((size_t)tw * p.wR + n) * p.ldw + c];
    } else {
#pragma unroll
      for (int n = 0; n < 4; ++n)
        if (n < p.N && n < p.ldw) v[n] = p.w[((size_t)tw * p.wR + c) * p.ldw + n];
    }
    *reinterpret_cast<f32x4*>(wl + e * 4) = v;
  }
  __syncthreads();
  const int q = lane % LPP, pl = lane / LPP;
  // grid-stride over groups of PPW pixels per wave: the weight table above is staged once per block, not once per 4 * PPW
  // pixels (the one-group-per-wave form ran the D_NET256 image gradient at 0.2 TB/s)
  const int ngroups = (p.M + PPW - 1) / PPW;
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int m = grp * PPW + pl;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int b = 0, oy = 0, ox = 0;
    if (m < p.M) {
      b = m >> p.lgHoWo;
      const int r = m & ((1 << p.lgHoWo) - 1);
      oy = r >> p.lgWo;
      ox = r & (p.Wo - 1);
      const int by = oy * s - pad, bx = ox * s - pad;
      const unsigned mask = tap_mask(p.kind, kw, by, bx, p.H, p.W, py, px);
      const long long xo = (((long long)b * p.H + by) * p.W + bx) * p.Cx + q * 4;
      for (int t = 0; t < p.T; ++t) {
        if (!((mask >> t) & 1u)) continue;
        int dy, dx;
        tap_delta(p.kind, kw, t, py, px, dy, dx);
        const long long xe = xo + ((long long)dy * p.W + dx) * p.Cx;
        f32x4 xv;
        if (p.x16) {
          const u32x2_t h = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned short*>(p.x) + xe);
          xv = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                     __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
        } else {
          xv = *reinterpret_cast<const f32x4*>(p.x + xe);
        }
        const float* wp = wl + ((size_t)t * p.Ca + q * 4) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc += xv[j] * *reinterpret_cast<const f32x4*>(wp + j * 4);
      }
    }
#pragma unroll
    for (int sft = 1; sft < LPP; sft <<= 1)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] += __shfl_xor(acc[n], sft);
    if (q == 0 && m < p.M) {
      long long row = m;
      if (p.kind == S2I_TCONV_K4S2) row = ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
      f32x4 o;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float v = acc[n];
        if (p.bias && n < p.N) v += p.bias[n];
        if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        else if (p.act == S2I_ACT_TANH) v = tanhf(v);
        else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
        o[n] = v;
      }
      if (p.N == 4 && !p.y16 && (p.ldy & 3) == 0) {
        *reinterpret_cast<f32x4*>(p.y + row * p.ldy) = o;
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if (n >= p.N) break;
          if (p.y16) reinterpret_cast<unsigned short*>(p.y)[row * p.ldy + n] = f2bf(o[n]);
          else p.y[row * p.ldy + n] = o[n];
        }
      }
    }
  }
}

// ---- thin layers: 3 (4) channels on one side --------------------------------------------------------------------------
// GET_IMAGE_G's conv3x3 -> RGB and the input gradient of the discriminators' first conv (few OUTPUT channels), the first
// discriminator conv itself and GET_IMAGE_G's input gradient (4 INPUT channels): ONE LANE PER OUTPUT PIXEL on the vector
// units, the weights as wave-uniform operands (scalar loads of a [phase][tap][k][n] fp32 table prepared by
// thin_table_kernel), so an FMA needs no LDS read and no cross-lane reduction.  HBM-bound by design; the 32-wide matrix
// tiles are 7/8 padding here and the lanes-per-pixel kernel above spends most of its time in LDS weight reads and shuffles.
__global__ void thin_table_kernel(const float* __restrict__ P, float* __restrict__ table, int kind, int flip, int T, int wt,
                                  int wR, int ldw, int Kk, int Nn, int nphases) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = nphases * T * Kk * Nn;
  if (e >= total) return;
  const int n = e % Nn, k = (e / Nn) % Kk, t = (e / (Nn * Kk)) % T, ph = e / (Nn * Kk * T);
  const int tw = tap_weight(kind, flip, T, t, ph >> 1, ph & 1);
  float v = 0.f;
  if (wt) { if (n < wR && k < ldw) v = P[((size_t)tw * wR + n) * ldw + k]; }
  else { if (k < wR && n < ldw) v = P[((size_t)tw * wR + k) * ldw + n]; }
  table[e] = v;
}

__device__ __forceinline__ void load8(const IgemmP& p, long long xe, float (&v)[8]) {
  if (p.x16) {
    const u32x4 h = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.x) + xe);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __builtin_bit_cast(float, h[j] << 16);
      v[2 * j + 1] = __builtin_bit_cast(float, h[j] & 0xffff0000u);
    }
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.x + xe), b = *reinterpret_cast<const f32x4*>(p.x + xe + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
  }
}

// <= 4 output channels: y[pix][0..3] = act(sum_{t,c} x[pix + t][c] * table[phase][t][c][0..3] + bias)
__global__ __launch_bounds__(256) void thin_out_kernel(IgemmP p, const float* __restrict__ table) {
  const int phase = blockIdx.z, py = phase >> 1, px = phase & 1;
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);
  const float* __restrict__ wph = table + (size_t)phase * p.T * p.Ca * 4;
  for (int m = blockIdx.x * 256 + threadIdx.x; m < p.M; m += gridDim.x * 256) {
    const int b = m >> p.lgHoWo;
    const int r = m & ((1 << p.lgHoWo) - 1);
    const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
    const int by = oy * s - pad, bx = ox * s - pad;
    const unsigned mask = tap_mask(p.kind, kw, by, bx, p.H, p.W, py, px);
    const long long xo = (((long long)b * p.H + by) * p.W + bx) * p.Cx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int t = 0; t < p.T; ++t) {
      if (!((mask >> t) & 1u)) continue;
      int dy, dx;
      tap_delta(p.kind, kw, t, py, px, dy, dx);
      const long long xe = xo + ((long long)dy * p.W + dx) * p.Cx;
      for (int c0 = 0; c0 < p.Ca; c0 += 8) {
        float v[8];
        load8(p, xe + c0, v);
        const float* __restrict__ w = wph + ((size_t)t * p.Ca + c0) * 4;   // wave-uniform
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a0 = fmaf(v[j], w[j * 4 + 0], a0);
          a1 = fmaf(v[j], w[j * 4 + 1], a1);
          a2 = fmaf(v[j], w[j * 4 + 2], a2);
          a3 = fmaf(v[j], w[j * 4 + 3], a3);
        }
      }
    }
    long long row = m;
    if (p.kind == S2I_TCONV_K4S2) row = ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
    f32x4 o = {a0, a1, a2, a3};
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float v = o[n];
      if (p.bias && n < p.N) v += p.bias[n];
      if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
      else if (p.act == S2I_ACT_TANH) v = tanhf(v);
      else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
      o[n] = v;
    }
    if (p.N == 4 && !p.y16 && (p.ldy & 3) == 0) {
      *reinterpret_cast<f32x4*>(p.y + row * p.ldy) = o;
    } else {
      for (int n = 0; n < p.N && n < 4; ++n) {
        if (p.y16) reinterpret_cast<unsigned short*>(p.y)[row * p.ldy + n] = f2bf(o[n]);
        else p.y[row * p.ldy + n] = o[n];
      }
    }
  }
}

// ---- transposed conv to <= 4 channels from 64 (the image gradient of the discriminators' first conv, model.py:383) ------
// The lanes-per-pixel kernel above ran this layer at 12 TFLOP/s-equivalent (0.27 ms for a 125 MB stream: every input pixel is
// 256 bytes and four lanes-per-pixel groups re-read it per output phase).  Here a block owns 8 x 8 INPUT pixels (+ 1 halo):
// the 10 x 10 x C patch is staged in LDS once with coalesced 16-byte loads, each of the four waves computes the 8 x 8 outputs
// of ONE output phase (py, px), so its 2 x 2 taps' weights are wave-uniform and arrive as scalar loads from the
// [phase][tap][c][4] table of thin_table_kernel; a lane reads its pixel's channels from LDS as 16-byte pieces (rows padded
// to C + 4 floats: the 16 lanes of a read group hit distinct bank groups).  HBM-bound by construction: x read once, y written
// once.
template <int C>
__global__ __launch_bounds__(256) void tconv_n4_tile_kernel(IgemmP p, const float* __restrict__ table) {
  constexpr int LDP = C + 4;                       // floats per patch pixel in LDS
  __shared__ __attribute__((aligned(16))) float patch[100 * LDP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int phase = __builtin_amdgcn_readfirstlane(tid >> 6), py = phase >> 1, px = phase & 1;
  const int tilesX = p.W >> 3, tilesY = p.H >> 3;
  const int tix = blockIdx.x % tilesX, tiy = (blockIdx.x / tilesX) % tilesY, b = blockIdx.x / (tilesX * tilesY);
  const int iy0 = tiy * 8 - 1, ix0 = tix * 8 - 1;
  // stage the patch: 100 pixels x C/4 float4 pieces
  for (int e = tid; e < 100 * (C / 4); e += 256) {
    const int pix = e / (C / 4), q = e - pix * (C / 4);
    const int yl = pix / 10, xl = pix - yl * 10;
    const int iy = iy0 + yl, ix = ix0 + xl;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
      const long long xe = (((long long)b * p.H + iy) * p.W + ix) * p.Cx + q * 4;
      if (p.x16) {
        const u32x2_t h = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned short*>(p.x) + xe);
        v = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                  __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
      } else {
        v = *reinterpret_cast<const f32x4*>(p.x + xe);
      }
    }
    *reinterpret_cast<f32x4*>(patch + pix * LDP + q * 4) = v;
  }
  __syncthreads();
  const int ly = lane >> 3, lx = lane & 7;                        // this lane's input pixel inside the tile
  const float* __restrict__ wph = table + (size_t)phase * 4 * C * 4;   // [tap][c][4], wave-uniform
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int dy, dx;
    tap_delta(S2I_TCONV_K4S2, 1, t, py, px, dy, dx);
    const float* xp = patch + ((ly + 1 + dy) * 10 + (lx + 1 + dx)) * LDP;
    const float* __restrict__ w = wph + (size_t)t * C * 4;
#pragma unroll 4
    for (int c0 = 0; c0 < C; c0 += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + c0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = fmaf(v[j], w[(c0 + j) * 4 + 0], a0);
        a1 = fmaf(v[j], w[(c0 + j) * 4 + 1], a1);
        a2 = fmaf(v[j], w[(c0 + j) * 4 + 2], a2);
        a3 = fmaf(v[j], w[(c0 + j) * 4 + 3], a3);
      }
    }
  }
  const int oy = 2 * (tiy * 8 + ly) + py, ox = 2 * (tix * 8 + lx) + px;
  const long long row = ((long long)b * (2 * p.H) + oy) * (2 * p.W) + ox;
  f32x4 o = {a0, a1, a2, a3};
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float v = o[n];
    if (p.bias && n < p.N) v += p.bias[n];
    if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
    else if (p.act == S2I_ACT_TANH) v = tanhf(v);
    else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
    o[n] = v;
  }
  if (p.N == 4 && !p.y16 && (p.ldy & 3) == 0) {
    *reinterpret_cast<f32x4*>(p.y + row * p.ldy) = o;
  } else {
    for (int n = 0; n < p.N && n < 4; ++n) {
      if (p.y16) reinterpret_cast<unsigned short*>(p.y)[row * p.ldy + n] = f2bf(o[n]);
      else p.y[row * p.ldy + n] = o[n];
    }
  }
}

// conv3x3 to <= 4 channels (GET_IMAGE_G, model.py:287-298) from 16 / 32 / 64 channels, the same construction: a block owns
// 16 x 16 output pixels, stages the 18 x 18 patch of (up to) 32 channels in LDS with coalesced 16-byte loads, one lane per
// output pixel, all nine taps' weights wave-uniform from the [tap][c][4] table.  x read once (+ 27 % halo), y written once.
template <int CCH>
__global__ __launch_bounds__(256) void conv3_n4_tile_kernel(IgemmP p, const float* __restrict__ table) {
  constexpr int LDP = CCH + 4;
  __shared__ __attribute__((aligned(16))) float patch[324 * LDP];
  const int tid = threadIdx.x;
  const int tilesX = p.W >> 4, tilesY = p.H >> 4;
  const int tix = blockIdx.x % tilesX, tiy = (blockIdx.x / tilesX) % tilesY, b = blockIdx.x / (tilesX * tilesY);
  const int iy0 = tiy * 16 - 1, ix0 = tix * 16 - 1;
  const int ly = tid >> 4, lx = tid & 15;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int cb = 0; cb < p.Ca; cb += CCH) {
    if (cb) __syncthreads();
    for (int e = tid; e < 324 * (CCH / 4); e += 256) {
      const int pix = e / (CCH / 4), q = e - pix * (CCH / 4);
      const int yl = pix / 18, xl = pix - yl * 18;
      const int iy = iy0 + yl, ix = ix0 + xl;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
        const long long xe = (((long long)b * p.H + iy) * p.W + ix) * p.Cx + cb + q * 4;
        if (p.x16) {
          const u32x2_t h = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned short*>(p.x) + xe);
          v = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                    __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
        } else {
          v = *reinterpret_cast<const f32x4*>(p.x + xe);
        }
      }
      *reinterpret_cast<f32x4*>(patch + pix * LDP + q * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float* xp = patch + ((ly + t / 3) * 18 + lx + t % 3) * LDP;
      const float* __restrict__ w = table + ((size_t)t * p.Ca + cb) * 4;     // wave-uniform
#pragma unroll 4
      for (int c0 = 0; c0 < CCH; c0 += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xp + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = fmaf(v[j], w[(c0 + j) * 4 + 0], a0);
          a1 = fmaf(v[j], w[(c0 + j) * 4 + 1], a1);
          a2 = fmaf(v[j], w[(c0 + j) * 4 + 2], a2);
          a3 = fmaf(v[j], w[(c0 + j) * 4 + 3], a3);
        }
      }
    }
  }
  const long long row = ((long long)b * p.H + tiy * 16 + ly) * p.W + tix * 16 + lx;
  f32x4 o = {a0, a1, a2, a3};
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float v = o[n];
    if (p.bias && n < p.N) v += p.bias[n];
    if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
    else if (p.act == S2I_ACT_TANH) v = tanhf(v);
    else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
    o[n] = v;
  }
  if (p.N == 4 && !p.y16 && (p.ldy & 3) == 0) {
    *reinterpret_cast<f32x4*>(p.y + row * p.ldy) = o;
  } else {
    for (int n = 0; n < p.N && n < 4; ++n) {
      if (p.y16) reinterpret_cast<unsigned short*>(p.y)[row * p.ldy + n] = f2bf(o[n]);
      else p.y[row * p.ldy + n] = o[n];
    }
  }
}

// 4 input channels, NOUT outputs: y[pix][n] = act(sum_{t,ci} x[pix + t][ci] * table[t][ci][n])
template <int NOUT>
__global__ __launch_bounds__(256) void thin_in_kernel(IgemmP p, const float* __restrict__ table) {
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);
  for (int m = blockIdx.x * 256 + threadIdx.x; m < p.M; m += gridDim.x * 256) {
    const int b = m >> p.lgHoWo;
    const int r = m & ((1 << p.lgHoWo) - 1);
    const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
    const int by = oy * s - pad, bx = ox * s - pad;
    const unsigned mask = tap_mask(p.kind, kw, by, bx, p.H, p.W, 0, 0);
    const long long xo = (((long long)b * p.H + by) * p.W + bx) * 4;
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
    for (int t = 0; t < p.T; ++t) {
      if (!((mask >> t) & 1u)) continue;
      int dy, dx;
      tap_delta(p.kind, kw, t, 0, 0, dy, dx);
      const long long xe = xo + ((long long)dy * p.W + dx) * 4;
      f32x4 xv;
      if (p.x16) {
        const u32x2_t h = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned short*>(p.x) + xe);
        xv = f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                   __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
      } else {
        xv = *reinterpret_cast<const f32x4*>(p.x + xe);
      }
      const float* __restrict__ w = table + (size_t)t * 4 * NOUT;   // wave-uniform
#pragma unroll
      for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int n = 0; n < NOUT; ++n) acc[n] = fmaf(xv[ci], w[ci * NOUT + n], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
      float v = acc[n];
      if (p.bias) v += p.bias[n];
      if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
      else if (p.act == S2I_ACT_TANH) v = tanhf(v);
      else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
      acc[n] = v;
    }
    if (p.y16) {
      unsigned short* yp = reinterpret_cast<unsigned short*>(p.y) + (long long)m * p.ldy;
#pragma unroll
      for (int g = 0; g < NOUT / 8; ++g) {
        u32x4 o;
#pragma unroll
        for (int h = 0; h < 4; ++h) o[h] = (unsigned)f2bf(acc[g * 8 + 2 * h]) | ((unsigned)f2bf(acc[g * 8 + 2 * h + 1]) << 16);
        *reinterpret_cast<u32x4*>(yp + g * 8) = o;
      }
    } else {
      float* yp = p.y + (long long)m * p.ldy;
#pragma unroll
      for (int g = 0; g < NOUT / 4; ++g)
        *reinterpret_cast<f32x4*>(yp + g * 4) = f32x4{acc[g * 4], acc[g * 4 + 1], acc[g * 4 + 2], acc[g * 4 + 3]};
    }
  }
}

// ---- bf16 mode: the 3-channel image layers on the bf16 matrix cores with PIXELS AS COLUMNS ---------------------------------
// D = W (32 output-channel rows x K) . X^T (K x 32 pixels): the weights are the A operand and stay in registers for the whole
// kernel (fragments prepared by rgb_afrag_kernel), the B fragment of a lane -- 8 consecutive K values of ITS pixel -- is 16
// (bf16) or 32 (fp32 NHWC4: two adjacent taps) contiguous bytes of global memory, so no LDS, no barriers and no cross-lane
// reduction; the result of a pixel sits in the registers of its own lane(s) and leaves as 8 / 16-byte stores.
//   rgb_out: few output channels (GET_IMAGE_G's conv3x3 -> RGB, input gradient of the first discriminator conv)
//   rgb_in : 4 input channels (first discriminator conv, input gradient of GET_IMAGE_G)
__global__ void rgb_afrag_kernel(const float* __restrict__ P, unsigned short* __restrict__ out, int mode, int kind, int flip, int T,
                                 int wt, int wR, int ldw, int CIN, int KW, int MT, int KSTEPS, int nphases, int Nreal) {
  // out[phase][mt][ks][lane][8]; mode 0 (rgb_out): k = t * CIN + c;  mode 1 (rgb_in): k-step = kernel row, j = dxl * 4 + c
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = nphases * MT * KSTEPS * 64 * 8;
  if (e >= total) return;
  const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) % KSTEPS, mt = ((e >> 9) / KSTEPS) % MT, ph = (e >> 9) / (KSTEPS * MT);
  const int n = mt * 32 + (lane & 31), kk = 8 * (lane >> 5) + j;
  int t, c;
  bool live = n < Nreal;
  if (mode == 0) {
    const int k = ks * 16 + kk;
    t = k / CIN;
    c = k - t * CIN;
  } else {
    const int dxl = kk >> 2;
    c = kk & 3;
    t = ks * KW + dxl;
    live = live && dxl < KW;
  }
  float v = 0.f;
  if (live) {
    const int tw = tap_weight(kind, flip, T, t, ph >> 1, ph & 1);
    if (wt) { if (n < wR && c < ldw) v = P[((size_t)tw * wR + n) * ldw + c]; }
    else { if (c < wR && n < ldw) v = P[((size_t)tw * wR + c) * ldw + n]; }
  }
  out[e] = f2bf(v);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int KSTEPS>
__global__ __launch_bounds__(256) void rgb_out_kernel(IgemmP p, const unsigned short* __restrict__ afrag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int phase = blockIdx.z, py = phase >> 1, px = phase & 1;
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);
  bf16x8_t A[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks)
    A[ks] = *reinterpret_cast<const bf16x8_t*>(afrag + ((size_t)(phase * KSTEPS + ks) * 64 + lane) * 8);
  const int kpt = p.Ca / 16;                      // k-steps per tap
  const unsigned short* xb = reinterpret_cast<const unsigned short*>(p.x);
  const int ngroups = (p.M + 31) / 32;
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int m = grp * 32 + l31;
    const bool live = m < p.M;
    const int b = m >> p.lgHoWo;
    const int r = m & ((1 << p.lgHoWo) - 1);
    const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
    const int by = oy * s - pad, bx = ox * s - pad;
    const unsigned mask = live ? tap_mask(p.kind, kw, by, bx, p.H, p.W, py, px) : 0u;
    const long long xo = (((long long)b * p.H + by) * p.W + bx) * p.Cx + 8 * lh;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int t = ks / kpt, c0 = (ks - t * kpt) * 16;
      int dy, dx;
      tap_delta(p.kind, kw, t, py, px, dy, dx);
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((mask >> t) & 1u) v = *reinterpret_cast<const u32x4*>(xb + xo + ((long long)dy * p.W + dx) * p.Cx + c0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ks], __builtin_bit_cast(bf16x8_t, v), acc, 0, 0, 0);
    }
    if (lh == 0 && live) {                       // rows 0..3 of column l31 = registers 0..3 of this lane
      long long row = m;
      if (p.kind == S2I_TCONV_K4S2) row = ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
      f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        float v = o[n];
        if (p.bias && n < p.N) v += p.bias[n];
        if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        else if (p.act == S2I_ACT_TANH) v = tanhf(v);
        else if (p.act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
        o[n] = v;
      }
      if (p.N == 4 && (p.ldy & 3) == 0) *reinterpret_cast<f32x4*>(p.y + row * p.ldy) = o;
      else
        for (int n = 0; n < p.N && n < 4; ++n) p.y[row * p.ldy + n] = o[n];
    }
  }
}

// MT = output-channel tiles of 32; KH = kernel rows = k-steps (a k-step holds the 4 horizontal taps x 4 channels of one row;
// the 3x3 has a zero fourth tap)
template <int MT, int KH>
__global__ __launch_bounds__(256) void rgb_in_kernel(IgemmP p, const unsigned short* __restrict__ afrag) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  int s, pad, kw;
  geom(p, p.kind, s, pad, kw);
  bf16x8_t A[MT][KH];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
      A[mt][ks] = *reinterpret_cast<const bf16x8_t*>(afrag + ((size_t)(mt * KH + ks) * 64 + lane) * 8);
  unsigned short* yb = reinterpret_cast<unsigned short*>(p.y);
  const int ngroups = (p.M + 31) / 32;
  for (int grp = blockIdx.x * 4 + wave; grp < ngroups; grp += gridDim.x * 4) {
    const int m = grp * 32 + l31;
    const bool live = m < p.M;
    const int b = m >> p.lgHoWo;
    const int r = m & ((1 << p.lgHoWo) - 1);
    const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
    const int by = oy * s - pad, bx = ox * s - pad + 2 * lh;   // this lane's two taps: columns bx, bx + 1
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mt][q] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      const int iy = by + ks;
      const bool rowok = live && iy >= 0 && iy < p.H;
      const float* xp = p.x + (((long long)b * p.H + iy) * p.W + bx) * 4;
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (rowok && bx >= 0 && bx < p.W) v0 = *reinterpret_cast<const f32x4*>(xp);
      if (rowok && bx + 1 >= 0 && bx + 1 < p.W) v1 = *reinterpret_cast<const f32x4*>(xp + 4);
      bf16x8_t bv = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3],
                     (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mt][ks], bv, acc[mt], 0, 0, 0);
    }
    if (!live) continue;
    // column l31 (this pixel): registers 4g..4g+3 of tile mt = channels mt*32 + 8g + 4lh + (0..3)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = mt * 32 + 8 * g + 4 * lh;
        if (n >= p.N) continue;
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = acc[mt][4 * g + q];
          if (p.act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
          else if (p.act == S2I_ACT_TANH) v = tanhf(v);
          o[q] = v;
        }
        *reinterpret_cast<u32x2_t*>(yb + (long long)m * p.ldy + n) =
            u32x2_t{(unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16), (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16)};
      }
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int S, long long rows, int N,
                                     const float* __restrict__ bias, int act, float* __restrict__ y,
                                     int ldy, int y16) {
  const long long total = rows * N;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long row = e / N;
    const int n = (int)(e - row * N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += slab[(size_t)s * total + e];
    if (bias) v += bias[n];
    if (act == S2I_ACT_LRELU) v = v > 0.f ? v : 0.2f * v;
    else if (act == S2I_ACT_TANH) v = tanhf(v);
    else if (act == S2I_ACT_RELU) v = fmaxf(v, 0.f);
    if (y16) reinterpret_cast<unsigned short*>(y)[row * ldy + n] = f2bf(v);
    else y[row * ldy + n] = v;
  }
}

// split-K reduction fused with the BatchNorm column statistics: y = sum_s slab[s], part = per-row-chunk column
// sums and sums of squares (the layout s2i_colstats writes).  256 threads = cpb column quads x 256/cpb row lanes.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ slab, int S, long long rows,
                                                                  int N, float* __restrict__ y, int ldy,
                                                                  float* __restrict__ part, int nparts, int cpb,
                                                                  int ppg, long long Rg, int y16) {
  __shared__ f32x4 sh[2][256];
  const int tid = threadIdx.x;
  const int rpb = 256 / cpb;
  const int ql = tid % cpb, rl = tid / cpb;
  const int quad = blockIdx.y * cpb + ql;
  const int Q = N / 4;
  const int grp = blockIdx.x / ppg, pp = blockIdx.x - grp * ppg;  // BatchNorm group of this row chunk
  const long long chunk = (Rg + ppg - 1) / ppg;
  const long long r0 = grp * Rg + pp * chunk;
  const long long gend = (grp + 1) * Rg < rows ? (grp + 1) * Rg : rows;
  const long long r1 = r0 + chunk < gend ? r0 + chunk : gend;
  const size_t sstride = (size_t)rows * N;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (quad < Q) {
    for (long long row = r0 + rl; row < r1; row += rpb) {
      const float* sp = slab + row * N + quad * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(sp);
      for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(sp + s * sstride);
      if (y16) {
        unsigned short* yp = reinterpret_cast<unsigned short*>(y) + row * ldy + quad * 4;
        *reinterpret_cast<u32x2_t*>(yp) = u32x2_t{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16),
                                                 (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
      } else {
        *reinterpret_cast<f32x4*>(y + row * ldy + quad * 4) = v;
      }
      s0 += v;
      s1 += v * v;
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
    *reinterpret_cast<f32x4*>(part + ((size_t)0 * nparts + blockIdx.x) * N + quad * 4) = s0;
    *reinterpret_cast<f32x4*>(part + ((size_t)1 * nparts + blockIdx.x) * N + quad * 4) = s1;
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: slab[split][krow][n] = sum_{pixels in split} A(pixel, krow) * g[pixel][n]
struct WgradP {
  const float* __restrict__ a;
  const float* __restrict__ cvec;
  const float* __restrict__ g;
  float* __restrict__ slab;
  int B, H, W, Ca, Cc, Cin;
  int Ho, Wo, lgWo, lgHoWo;
  int M, N, ldg, K, T, kind;
  int cps, nchunks;
  int a16, g16;  // a / g hold bf16 instead of fp32
  unsigned a_bytes, c_bytes, g_bytes;
  // apply-on-load (AACT instantiation): `a` holds the RAW output of the producing convolution, a_coef its (groups, 4, Ca)
  // BatchNorm coefficient table; the gather computes LeakyReLU(scale * a + shift), padding taps staying zero
  const float* __restrict__ a_coef;
  int a_groups, a_ipg;   // BatchNorm groups of the producer, images per group
};

// XCD-aware block order of the weight-gradient grids (tiles x pixel-range splits).  Every tile of ONE split reads the same
// pixels of both operands, and blocks are dealt round-robin over the chip's 8 XCDs, each with an L2 of its own
// (MI355X_MICROARCH.md): in launch order (tile fastest) the 8 k-tiles of a split land on 8 different XCDs and every one of
// them pulls the split's `g` rows through the fabric (profiles/r03_roofline_bf16_wgrad_b48: 1.25 GB fetched for 453 MB of
// operands).  Here linear block id L maps to (tile, split) such that all tiles of a split have the same L % 8 -- one XCD --
// and consecutive ids on that XCD; a bijection for any split count (the last group of splits uses its own modulus).
__device__ __forceinline__ void wgrad_block_map(int& tile, int& split) {
  const int tiles = gridDim.x * gridDim.y, nsplit = gridDim.z;
  const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const int per_group = 8 * tiles;
  const int grp = L / per_group, Ll = L - grp * per_group;
  const int in_group = min(8, nsplit - grp * 8);         // splits of this group (the last one may hold fewer)
  split = grp * 8 + Ll % in_group;
  tile = Ll / in_group;
}

// (HIP's second launch bound is waves per SIMD: 3 = three 256-thread blocks per CU, 4 = two 512-thread blocks or one of 1024)
template <int BM, int BN, int WAVES_M, int WAVES_N, bool AACT = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N > 4 ? 4 : 3)) void igemm_wgrad_kernel(WgradP p) {
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int LDA = BM, LDB = BN;
  constexpr int NT = 64 * WAVES_M * WAVES_N;  // 256, or 192 for the 96-row tiles (K = 9 * 32)
  constexpr int AROWS = NT * 4 / BM, BROWS = NT * 4 / BN;  // pixel rows of the 32-deep chunk staged per pass
  static_assert(AROWS * BM == NT * 4 && BROWS * BN == NT * 4, "a pass must cover whole rows");
  constexpr int APASS = (32 + AROWS - 1) / AROWS, BPASS = (32 + BROWS - 1) / BROWS;
  constexpr bool APRED = (32 % AROWS) != 0, BPRED = (32 % BROWS) != 0;  // last pass partly beyond the chunk
  __shared__ __attribute__((aligned(16))) float smem[32 * LDA + 32 * LDB];
  float* As = smem;
  float* Bs = smem + 32 * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_, split;
  wgrad_block_map(tile_, split);
  const int k0 = (tile_ % gridDim.x) * BM, n0 = (tile_ / gridDim.x) * BN;
  int s, pad, kw;
  geom(p.kind, s, pad, kw);

  const int acol4 = tid % (BM / 4), arow = tid / (BM / 4);
  const int bcol4 = tid % (BN / 4), brow = tid / (BN / 4);
  // this thread's 4 gathered columns: fixed (tap, channel) for the whole pixel loop
  const int kcol = k0 + acol4 * 4;
  const bool kvalid = kcol < p.K;
  int c = 0, dy = 0, dx = 0;
  if (kvalid) {
    const int t = kcol / p.Cin;
    c = kcol - t * p.Cin;
    dy = t / kw;
    dx = t - dy * kw;
  }
  const bool from_vec = c < p.Cc;
  const int nb = n0 + bcol4 * 4;
  const bool nvalid = nb < p.N;

  const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.cvec, 0, p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
  const int acolb = (c - p.Cc) * 4;            // byte offset of this thread's 4 channels inside a pixel of a
  const int ccolb = c * 4;                     // ... inside a row of cvec
  const int gcolb = nvalid ? nb * 4 : S2I_OOB;
  f32x4 ra[APASS], rb[BPASS];
  // apply-on-load: this thread's four channels are the same for the whole pixel loop, so their scale / shift (per producer
  // group: at most three, the stacked real / wrong / fake passes) sit in registers; per pass, which group the pixel's image
  // belongs to (2 bits) and whether the tap is inside the image (1 bit)
  f32x4 gs[3], gt[3];
  unsigned apass = 0;
  if constexpr (AACT) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gg = g < p.a_groups ? g : 0;
      gs[g] = kvalid ? *reinterpret_cast<const f32x4*>(p.a_coef + ((size_t)gg * 4 + 2) * p.Ca + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      gt[g] = kvalid ? *reinterpret_cast<const f32x4*>(p.a_coef + ((size_t)gg * 4 + 3) * p.Ca + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  auto fetch = [&](int pc) {
    if constexpr (AACT) apass = 0;
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      const int m = pc * 32 + arow + q * AROWS;
      if (APRED && arow + q * AROWS >= 32) continue;
      const int b = m >> p.lgHoWo;
      const int r = m & ((1 << p.lgHoWo) - 1);
      const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
      const int iy = oy * s - pad + dy, ix = ox * s - pad + dx;
      const bool ok = kvalid && m < p.M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      if constexpr (AACT) {
        const unsigned grp = (unsigned)(b >= p.a_ipg) + (unsigned)(b >= 2 * p.a_ipg);
        apass |= ((ok ? 4u : 0u) | grp) << (3 * q);
      }
      if (from_vec) ra[q] = bload4(rc_rs, ok ? b * p.Cc * 4 + ccolb : S2I_OOB);
      else ra[q] = bload4_any(ra_rs, ok ? ((b * p.H + iy) * p.W + ix) * p.Ca * 4 + acolb : S2I_OOB, p.a16);
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int m = pc * 32 + brow + q * BROWS;
      if (BPRED && brow + q * BROWS >= 32) continue;
      rb[q] = bload4_any(rg_rs, (m < p.M && nvalid) ? m * p.ldg * 4 + gcolb : S2I_OOB, p.g16);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunks, c_begin + p.cps);
  if (c_begin < c_end) fetch(c_begin);
  for (int pc = c_begin; pc < c_end; ++pc) {
#pragma unroll
    for (int q = 0; q < APASS; ++q)
      if (!APRED || arow + q * AROWS < 32) {
        f32x4 v = ra[q];
        if constexpr (AACT) {
          const unsigned bits = (apass >> (3 * q)) & 7u;
          const unsigned grp = bits & 3u;
          const f32x4 sc = grp == 0 ? gs[0] : (grp == 1 ? gs[1] : gs[2]);
          const f32x4 sh = grp == 0 ? gt[0] : (grp == 1 ? gt[1] : gt[2]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float z = fmaf(v[j], sc[j], sh[j]);
            z = z > 0.f ? z : 0.2f * z;
            v[j] = (bits & 4u) ? z : 0.f;
          }
        }
        *reinterpret_cast<f32x4*>(As + (arow + q * AROWS) * LDA + acol4 * 4) = v;
      }
#pragma unroll
    for (int q = 0; q < BPASS; ++q)
      if (!BPRED || brow + q * BROWS < 32) *reinterpret_cast<f32x4*>(Bs + (brow + q * BROWS) * LDB + bcol4 * 4) = rb[q];
    __syncthreads();
    if (pc + 1 < c_end) fetch(pc + 1);
    mma_chunk<TM, TN, LDA, LDB>(As, Bs, wm * TM * 32, wn * TN * 32, lane, acc);
    __syncthreads();
  }

  const int l31 = lane & 31, lh = lane >> 5;
  float* outp = p.slab + (size_t)split * p.K * p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = k0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (krow >= p.K) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n < p.N) outp[(size_t)krow * p.N + n] = acc[i][j][r];
      }
    }
}

// Split-bf16 weight gradient (see igemm_fwd_split_kernel).  The reduction index of this GEMM is the pixel, and both
// operands arrive pixel-major (NHWC), so the LDS images stay [pixel][row] -- a staged float4 becomes one 8-byte write per
// plane -- and the MFMA fragments (8 consecutive PIXELS of one row per lane) are fetched with the transposing read
// ds_read_b64_tr_b16: per 16-lane group it takes a 4-pixel x 16-row block and hands lane i column i.  16-byte chunks
// of a pixel row are XOR-swizzled with ((pixel & 3) << 2) | ((pixel >> 2) & 3) (cdna_hip_programming.md T10, image (b)).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ s16x4 lds_tr_read(const unsigned char* ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ptr));
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NP>
__global__ __launch_bounds__(256, 3) void igemm_wgrad_split_kernel(WgradP p) {
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int AROWB = BM * 2, BROWB = BN * 2;
  constexpr int APLANE = 32 * AROWB, BPLANE = 32 * BROWB;
  constexpr int AMASK = BM / 8 - 1, BMASK = BN / 8 - 1;
  constexpr int APASS = BM / 32, BPASS = BN / 32;
  constexpr int AROWS = 1024 / BM, BROWS = 1024 / BN;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NP * (APLANE + BPLANE)];
  unsigned char* As = smem;
  unsigned char* Bs = smem + NP * APLANE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int k0 = blockIdx.x * BM, n0 = blockIdx.y * BN, split = blockIdx.z;
  int s, pad, kw;
  geom(p.kind, s, pad, kw);

  const int acol4 = tid % (BM / 4), arow = tid / (BM / 4);
  const int bcol4 = tid % (BN / 4), brow = tid / (BN / 4);
  const int kcol = k0 + acol4 * 4;
  const bool kvalid = kcol < p.K;
  int c = 0, dy = 0, dx = 0;
  if (kvalid) {
    const int t = kcol / p.Cin;
    c = kcol - t * p.Cin;
    dy = t / kw;
    dx = t - dy * kw;
  }
  const bool from_vec = c < p.Cc;
  const int nb = n0 + bcol4 * 4;
  const bool nvalid = nb < p.N;
  const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.cvec, 0, p.c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
  const int acolb = (c - p.Cc) * 4;
  const int ccolb = c * 4;
  const int gcolb = nvalid ? nb * 4 : S2I_OOB;
  f32x4 ra[APASS], rb[BPASS];
  auto fetch = [&](int pc) {
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      const int m = pc * 32 + arow + q * AROWS;
      const int b = m >> p.lgHoWo;
      const int r = m & ((1 << p.lgHoWo) - 1);
      const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
      const int iy = oy * s - pad + dy, ix = ox * s - pad + dx;
      const bool ok = kvalid && m < p.M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      if (from_vec) ra[q] = bload4(rc_rs, ok ? b * p.Cc * 4 + ccolb : S2I_OOB);
      else ra[q] = bload4(ra_rs, ok ? ((b * p.H + iy) * p.W + ix) * p.Ca * 4 + acolb : S2I_OOB);
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int m = pc * 32 + brow + q * BROWS;
      rb[q] = bload4(rg_rs, (m < p.M && nvalid) ? m * p.ldg * 4 + gcolb : S2I_OOB);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing of this lane (see header): group g = lane >> 4 -> (h = g >> 1, 16-row block g & 1)
  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3;
  const int gh = lane >> 5, gcb = (lane >> 4) & 1;
  const int sw0 = (gq << 2) | (2 * gh);              // swizzle of pixel rows 8h + q (+ 16 ks); rows + 4: sw0 | 1
  int aad[TM][2], bad[TN][2];                        // byte offsets inside a plane for the two 4-pixel half fragments
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int prow = 8 * gh + 4 * f + gq;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ch = ((wm * TM + i) * 32 + 16 * gcb) / 8 + (gp >> 1);
      aad[i][f] = prow * AROWB + 16 * ((ch ^ (sw0 | f)) & AMASK) + 8 * (gp & 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ch = ((wn * TN + j) * 32 + 16 * gcb) / 8 + (gp >> 1);
      bad[j][f] = prow * BROWB + 16 * ((ch ^ (sw0 | f)) & BMASK) + 8 * (gp & 1);
    }
  }

  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunks, c_begin + p.cps);
  if (c_begin < c_end) fetch(c_begin);
  for (int pc = c_begin; pc < c_end; ++pc) {
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      const int row = arow + q * AROWS;
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
      u32x2 sp[NP];
      split4<NP>(ra[q], sp);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        *reinterpret_cast<u32x2*>(As + pl * APLANE + row * AROWB + 16 * (((acol4 >> 1) ^ sw) & AMASK) + 8 * (acol4 & 1)) = sp[pl];
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int row = brow + q * BROWS;
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
      u32x2 sp[NP];
      split4<NP>(rb[q], sp);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        *reinterpret_cast<u32x2*>(Bs + pl * BPLANE + row * BROWB + 16 * (((bcol4 >> 1) ^ sw) & BMASK) + 8 * (bcol4 & 1)) = sp[pl];
    }
    if (pc + 1 < c_end) fetch(pc + 1);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[NP][TM];
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const s16x4 lo = lds_tr_read(As + pl * APLANE + ks * 16 * AROWB + aad[i][0]);
          const s16x4 hi = lds_tr_read(As + pl * APLANE + ks * 16 * AROWB + aad[i][1]);
          a[pl][i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
      for (int pb = NP - 1; pb >= 0; --pb) {
        bf16x8 b[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const s16x4 lo = lds_tr_read(Bs + pb * BPLANE + ks * 16 * BROWB + bad[j][0]);
          const s16x4 hi = lds_tr_read(Bs + pb * BPLANE + ks * 16 * BROWB + bad[j][1]);
          b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int pa = NP - 1 - pb; pa >= 0; --pa)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int l31 = lane & 31, lh = lane >> 5;
  float* outp = p.slab + (size_t)split * p.K * p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = k0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (krow >= p.K) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n < p.N) outp[(size_t)krow * p.N + n] = acc[i][j][r];
      }
    }
}

// Weight gradient with BOTH operands stored as bf16 (bf16 activation mode): the structure of igemm_wgrad_split_kernel
// with one plane, but the staged values are already bf16, so a 16-byte load (8 channels of one pixel) is copied to
// LDS as it is, and a stage is 64 pixels deep (4 k-steps, 16 MFMAs per wave between two barriers instead of 8).
// A32: the gathered operand is the fp32 NHWC4 image of the first discriminator conv (4x4 stride 2): 8 consecutive K columns
// are two horizontally adjacent taps x 4 channels = 32 contiguous bytes, converted to bf16 while they are staged.
// 256 x 128 tiles (8 waves, two blocks per CU; round 3): a tile of BM x BN moves (BM + BN) * 2 bytes per pixel from L2 into LDS
// for 2 * BM * BN FLOP -- 64 FLOP/B at 128 x 128, which at the ~70 GB/s a CU takes from L2 (MI355X_MICROARCH.md, gather into
// LDS) caps the chip near 0.65 PFLOP/s, where the 128 x 128 form sat (profiles/r03_roofline_bf16_wgrad_b48); 85 FLOP/B here.
// 256 x 256 tiles on 1024-thread blocks (one per CU) where N allows: 128 FLOP/B, 0.89 PFLOP/s on D_NET256's deep layers against
// 0.80 (256 x 128) and 0.70 (128 x 128); two LDS stages with one barrier per stage measured the same and were removed.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool A32 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (WAVES_M * WAVES_N > 4 ? 4 : 3)) void igemm_wgrad_b16_kernel(WgradP p) {
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int PC = 64;                               // pixels per stage
  constexpr int AROWB = BM * 2, BROWB = BN * 2;
  constexpr int AMASK = BM / 8 - 1, BMASK = BN / 8 - 1;
  constexpr int ATPR = BM / 8, BTPR = BN / 8;          // threads per pixel row
  constexpr int AROWS = NT / ATPR, BROWS = NT / BTPR;  // pixel rows per pass
  static_assert(AROWS * ATPR == NT && BROWS * BTPR == NT && PC % AROWS == 0, "a pass covers whole pixel rows");
  constexpr int APASS = PC / AROWS, BPASS = (PC + BROWS - 1) / BROWS;
  constexpr int STAGE_BYTES = PC * (AROWB + BROWB);
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + PC * AROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  int tile_, split;
  wgrad_block_map(tile_, split);
  const int k0 = (tile_ % gridDim.x) * BM, n0 = (tile_ / gridDim.x) * BN;
  int s, pad, kw;
  geom(p.kind, s, pad, kw);
  const int acol8 = tid % ATPR, arow = tid / ATPR;
  const int bcol8 = tid % BTPR, brow = tid / BTPR;
  const int kcol = k0 + acol8 * 8;
  const bool kvalid = kcol < p.K;
  int c = 0, dy = 0, dx = 0;
  if (kvalid) {
    const int t = kcol / p.Cin;
    c = kcol - t * p.Cin;
    dy = t / kw;
    dx = t - dy * kw;
  }
  const int nb = n0 + bcol8 * 8;
  const bool nvalid = nb < p.N;
  const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);
  u32x4 ra[APASS], rb[BPASS];
  auto fetch = [&](int pc) {
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      const int m = pc * PC + arow + q * AROWS;
      const int b = m >> p.lgHoWo;
      const int r = m & ((1 << p.lgHoWo) - 1);
      const int oy = r >> p.lgWo, ox = r & (p.Wo - 1);
      const int iy = oy * s - pad + dy, ix = ox * s - pad + dx;
      if constexpr (A32) {
        const bool rowok = kvalid && m < p.M && iy >= 0 && iy < p.H;
        const int o0 = ((b * p.H + iy) * p.W + ix) * 16;      // byte offset of pixel (iy, ix): 4 fp32 channels
        const f32x4 v0 = bload4(ra_rs, (rowok && ix >= 0 && ix < p.W) ? o0 : S2I_OOB);
        const f32x4 v1 = bload4(ra_rs, (rowok && ix + 1 >= 0 && ix + 1 < p.W) ? o0 + 16 : S2I_OOB);
        ra[q] = u32x4{(unsigned)f2bf(v0[0]) | ((unsigned)f2bf(v0[1]) << 16), (unsigned)f2bf(v0[2]) | ((unsigned)f2bf(v0[3]) << 16),
                      (unsigned)f2bf(v1[0]) | ((unsigned)f2bf(v1[1]) << 16), (unsigned)f2bf(v1[2]) | ((unsigned)f2bf(v1[3]) << 16)};
      } else {
        const bool ok = kvalid && m < p.M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        ra[q] = __builtin_amdgcn_raw_buffer_load_b128(ra_rs, ok ? (((b * p.H + iy) * p.W + ix) * p.Ca + c) * 2 : S2I_OOB, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int row = brow + q * BROWS;
      const int m = pc * PC + row;
      rb[q] = __builtin_amdgcn_raw_buffer_load_b128(rg_rs, (row < PC && m < p.M && nvalid) ? (m * p.ldg + nb) * 2 : S2I_OOB, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int gi = lane & 15, gq = gi >> 2, gp = gi & 3;
  const int gh = lane >> 5, gcb = (lane >> 4) & 1;
  const int sw0 = (gq << 2) | (2 * gh);
  int aad[TM][2], bad[TN][2];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int prow = 8 * gh + 4 * f + gq;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ch = ((wm * TM + i) * 32 + 16 * gcb) / 8 + (gp >> 1);
      aad[i][f] = prow * AROWB + 16 * ((ch ^ (sw0 | f)) & AMASK) + 8 * (gp & 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ch = ((wn * TN + j) * 32 + 16 * gcb) / 8 + (gp >> 1);
      bad[j][f] = prow * BROWB + 16 * ((ch ^ (sw0 | f)) & BMASK) + 8 * (gp & 1);
    }
  }

  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunks, c_begin + p.cps);
  auto stage_store = [&](unsigned char* A_, unsigned char* B_) {
#pragma unroll
    for (int q = 0; q < APASS; ++q) {
      const int row = arow + q * AROWS;
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
      *reinterpret_cast<u32x4*>(A_ + row * AROWB + 16 * ((acol8 ^ sw) & AMASK)) = ra[q];
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int row = brow + q * BROWS;
      const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
      if (row < PC) *reinterpret_cast<u32x4*>(B_ + row * BROWB + 16 * ((bcol8 ^ sw) & BMASK)) = rb[q];
    }
  };
  auto stage_mma = [&](const unsigned char* A_, const unsigned char* B_) {
#pragma unroll
    for (int ks = 0; ks < PC / 16; ++ks) {
      bf16x8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const s16x4 lo = lds_tr_read(A_ + ks * 16 * AROWB + aad[i][0]);
        const s16x4 hi = lds_tr_read(A_ + ks * 16 * AROWB + aad[i][1]);
        a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const s16x4 lo = lds_tr_read(B_ + ks * 16 * BROWB + bad[j][0]);
        const s16x4 hi = lds_tr_read(B_ + ks * 16 * BROWB + bad[j][1]);
        b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };
  if (c_begin < c_end) fetch(c_begin);
  for (int pc = c_begin; pc < c_end; ++pc) {
    stage_store(As, Bs);
    __syncthreads();
    if (pc + 1 < c_end) fetch(pc + 1);
    stage_mma(As, Bs);
    __syncthreads();
  }

  const int l31 = lane & 31, lh = lane >> 5;
  float* outp = p.slab + (size_t)split * p.K * p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = k0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (krow >= p.K) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n < p.N) outp[(size_t)krow * p.N + n] = acc[i][j][r];
      }
    }
}

// Weight gradient of a 3x3 stride-1 convolution over a wide map with few channels (the generator at 64x64 and
// 128x128, Cin = 64 / 32).  The generic kernel above stages an im2col tile per chunk, i.e. it pulls every input pixel
// through the vector-memory path once per tap; with K x N this small that path, not the matrix cores, is the limit
// (57-78 TFLOP/s).  Here a block owns ONE kernel row dy and a chunk is 32 consecutive pixels of one image row: the
// block stages the 34-pixel input row segment (iy = y + dy - 1, halo of one pixel each side) ONCE and the three
// horizontal taps read their MFMA fragments from it at pixel offsets 0/1/2 -- a third of the loads, no wasted rows
// (block tile = (3 * CIN) x BN).  Slab rows are (tap, cin) as above, so the slab sum / OIHW finish are shared.
template <int CIN, int BN, int WAVES_M, int WAVES_N, int DYS>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 3) void wgrad_k3_rows_kernel(WgradP p) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int BM = 3 * CIN * DYS, RT = BM / 32;  // DYS = 3: the block owns all three kernel rows (Cin = 32)
  constexpr int TM = RT / WAVES_M, TN = BN / (32 * WAVES_N);
  static_assert(TM * WAVES_M == RT && TN * WAVES_N * 32 == BN, "tile split");
  constexpr int LDH = CIN, LDB = BN;
  constexpr int CQ = CIN / 4, HQR = 34 * CQ, HQ = DYS * HQR;  // float4 per staged row segment / in total
  constexpr int HPASS = (HQ + NT - 1) / NT;
  constexpr int BROWS = NT * 4 / BN, BPASS = (32 + BROWS - 1) / BROWS;
  static_assert(BROWS * BN == NT * 4, "a pass must cover whole rows");
  __shared__ __attribute__((aligned(16))) float smem[DYS * 34 * LDH + 32 * LDB];
  float* Hs = smem;
  float* Bs = smem + DYS * 34 * LDH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int dy0 = DYS == 3 ? 0 : blockIdx.x, n0 = blockIdx.y * BN, split = blockIdx.z;
  const int bcol4 = tid % (BN / 4), brow = tid / (BN / 4);
  const int nb = n0 + bcol4 * 4;
  const int gcolb = nb < p.N ? nb * 4 : S2I_OOB;
  const __amdgpu_buffer_rsrc_t ra_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.g, 0, p.g_bytes, 0x00020000);

  f32x4 rh[HPASS], rb[BPASS];
  auto fetch = [&](int pc) {
    const int m0 = pc * 32;
    const int b = m0 >> p.lgHoWo;
    const int r = m0 & ((1 << p.lgHoWo) - 1);
    const int y = r >> p.lgWo, x0 = r & (p.W - 1);
#pragma unroll
    for (int q = 0; q < HPASS; ++q) {
      const int e = tid + q * NT;
      if (e >= HQ) continue;
      const int dyl = e / HQR, er = e - dyl * HQR;
      const int iy = y + dy0 + dyl - 1;
      const int ix = x0 - 1 + er / CQ;
      const bool ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      rh[q] = bload4(ra_rs, ok ? ((b * p.H + iy) * p.W + x0 - 1) * CIN * 4 + er * 16 : S2I_OOB);
    }
#pragma unroll
    for (int q = 0; q < BPASS; ++q) {
      const int row = brow + q * BROWS;
      if (row >= 32) continue;
      rb[q] = bload4(rg_rs, gcolb == S2I_OOB ? S2I_OOB : (m0 + row) * p.ldg * 4 + gcolb);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, lh = lane >> 5;
  int aoff[TM];  // row tile -> (horizontal tap, channel half) -> offset inside the halo row segment
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rt = wm * TM + i;                 // row tile -> (kernel row, horizontal tap, channel half)
    const int dyl = rt / (3 * CIN / 32), rr = rt % (3 * CIN / 32);
    aoff[i] = dyl * 34 * LDH + (rr / (CIN / 32)) * LDH + (rr % (CIN / 32)) * 32;
  }
  const float* hp = Hs + lh * LDH + l31;
  const float* bp = Bs + lh * LDB + wn * TN * 32 + l31;

  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunks, c_begin + p.cps);
  if (c_begin < c_end) fetch(c_begin);
  for (int pc = c_begin; pc < c_end; ++pc) {
#pragma unroll
    for (int q = 0; q < HPASS; ++q)
      if (tid + q * NT < HQ) *reinterpret_cast<f32x4*>(Hs + (tid + q * NT) * 4) = rh[q];
#pragma unroll
    for (int q = 0; q < BPASS; ++q)
      if (brow + q * BROWS < 32) *reinterpret_cast<f32x4*>(Bs + (brow + q * BROWS) * LDB + bcol4 * 4) = rb[q];
    __syncthreads();
    if (pc + 1 < c_end) fetch(pc + 1);
    {
      float a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a0[i] = hp[aoff[i]];
#pragma unroll
      for (int j = 0; j < TN; ++j) b0[j] = bp[j * 32];
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a1[i] = hp[(2 * (kk + 1)) * LDH + aoff[i]];
#pragma unroll
        for (int j = 0; j < TN; ++j) b1[j] = bp[(2 * (kk + 1)) * LDB + j * 32];
        if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
        if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
        if (kk + 2 < 16) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a0[i] = hp[(2 * (kk + 2)) * LDH + aoff[i]];
#pragma unroll
          for (int j = 0; j < TN; ++j) b0[j] = bp[(2 * (kk + 2)) * LDB + j * 32];
        }
        if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
        if constexpr (TM * TN >= 4) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
  }

  float* outp = p.slab + (size_t)split * p.K * p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = dy0 * 3 * CIN + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;  // (dy*3 + dx) * CIN + c
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n < p.N) outp[(size_t)krow * p.N + n] = acc[i][j][r];
      }
    }
}

// Weight gradient of a 3x3 convolution with at most 4 output channels (GET_IMAGE_G's conv3x3 -> RGB, model.py:287-298):
// K x N = (9 * Ca) x 4 is far too small for matrix cores and the operands are read exactly once, so this is an
// HBM stream.  LPP = Ca/4 lanes share one INPUT pixel (a wave reads 64 consecutive float4 = 1 KB of NHWC), each lane
// multiplies its 4 channels with the float4 output gradient of the 9 output pixels that see this input pixel and
// keeps all 9 x 4 x 4 products in registers across its pixel loop.  One slab [9*Ca][4] per block.
template <int LPP>
__global__ __launch_bounds__(256) void small_n_wgrad_kernel(WgradP p) {
  constexpr int PPW = 64 / LPP;
  __shared__ f32x4 red[4][9 * LPP * 4];  // [wave][tap][q][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane % LPP, pl = lane / LPP;
  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ngroups = p.M / PPW;  // W is a power of two >= PPW, so a group never straddles an image row
  const int wmask = p.W - 1;
  // two pixel groups per trip: both groups' loads (their input quads and the 2 x 9 output-gradient pixels) are issued before
  // the first product, which doubles the bytes each wave keeps in flight -- the kernel is a latency-bound stream at two waves
  // per SIMD (144 accumulators), 147 us for 125 MB with one group per trip
  auto load_a = [&](int m) -> f32x4 {
    if (p.a16) {
      const u32x2_t h = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned short*>(p.a) + (size_t)m * p.Ca + q * 4);
      return f32x4{__builtin_bit_cast(float, h[0] << 16), __builtin_bit_cast(float, h[0] & 0xffff0000u),
                   __builtin_bit_cast(float, h[1] << 16), __builtin_bit_cast(float, h[1] & 0xffff0000u)};
    }
    return *reinterpret_cast<const f32x4*>(p.a + (size_t)m * p.Ca + q * 4);
  };
  auto load_g = [&](int m, f32x4 (&gv)[9]) {
    const int ix = m & wmask;
    const int iy = (m >> p.lgWo) & (p.H - 1);
    const float* gp = p.g + (size_t)m * 4;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int oy = iy + 1 - dy, ox = ix + 1 - dx;
        const bool ok = oy >= 0 && oy < p.H && ox >= 0 && ox < p.W;
        gv[dy * 3 + dx] = ok ? *reinterpret_cast<const f32x4*>(gp + ((1 - dy) * p.W + (1 - dx)) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  auto fma9 = [&](const f32x4& av, const f32x4 (&gv)[9]) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t][j] += av[j] * gv[t];
  };
  const int stride = gridDim.x * 4;
  int grp = blockIdx.x * 4 + wave;
  for (; grp + stride < ngroups; grp += 2 * stride) {
    const int m0 = grp * PPW + pl, m1 = (grp + stride) * PPW + pl;
    f32x4 g0[9], g1[9];
    const f32x4 av0 = load_a(m0), av1 = load_a(m1);
    load_g(m0, g0);
    load_g(m1, g1);
    fma9(av0, g0);
    fma9(av1, g1);
  }
  if (grp < ngroups) {
    const int m0 = grp * PPW + pl;
    f32x4 g0[9];
    const f32x4 av0 = load_a(m0);
    load_g(m0, g0);
    fma9(av0, g0);
  }
  // lanes with equal q (different pixels) -> lane q
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = acc[t][j];
#pragma unroll
      for (int sft = LPP; sft < 64; sft <<= 1)
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] += __shfl_xor(v[n], sft);
      if (pl == 0) red[wave][(t * LPP + q) * 4 + j] = v;
    }
  __syncthreads();
  float* outp = p.slab + (size_t)blockIdx.x * p.K * 4;
  for (int e = tid; e < 9 * LPP * 4; e += 256) {
    const f32x4 v = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    *reinterpret_cast<f32x4*>(outp + (size_t)e * 4) = v;  // e = tap * Ca + channel: the slab's K row
  }
}

// slab[0] = sum_s slab[s] for a SMALL K x N (n4 float4) and many slabs: one block per 4 float4, 64 slab lanes each
__global__ __launch_bounds__(256) void slab_sum_tree_kernel(float* __restrict__ slab, int S, int n4) {
  __shared__ f32x4 sh[256];
  const int tid = threadIdx.x;
  const int e = blockIdx.x * 4 + (tid & 3), sl = tid >> 2;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (e < n4)
    for (int s = sl; s < S; s += 64) v += *reinterpret_cast<const f32x4*>(slab + ((size_t)s * n4 + e) * 4);
  sh[tid] = v;
  __syncthreads();
  for (int h = 32; h >= 1; h >>= 1) {
    if (sl < h) sh[tid] += sh[tid + h * 4];
    __syncthreads();
  }
  if (sl == 0 && e < n4) *reinterpret_cast<f32x4*>(slab + (size_t)e * 4) = sh[tid];
}

// slab[0] += slab[1..S-1]: a pure float4 stream over the split slabs (full-chip parallel, HBM-bound)
__global__ __launch_bounds__(256) void slab_sum_kernel(float* __restrict__ slab, int S, long long n4) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4;
       e += (long long)gridDim.x * blockDim.x) {
    f32x4 v = *reinterpret_cast<const f32x4*>(slab + e * 4);
    for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(slab + ((size_t)s * n4 + e) * 4);
    *reinterpret_cast<f32x4*>(slab + e * 4) = v;
  }
}

// Reduce the split slabs and write the reference's OIHW gradient tensor.  A [T][RT][32] tile goes through
// LDS so that both the slab reads (32 consecutive columns) and the OIHW writes (runs of (cin, ky, kx) for one
// cout) are coalesced.  fold: the 3x3 parameter tap (ky,kx) collects the 4 effective 4x4 taps it was summed into.
//   non-swap: slab rows (tap, cin), columns cout;   swap: slab rows (tap, cout), columns cin.
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slab, int S, int K, int N,
                                                           int Cg, int O, int I, int Tp, int T, int swap, int fold,
                                                           int accumulate, int RT, float* __restrict__ grad,
                                                           int i_off, int I_total) {
  extern __shared__ float tile[];  // [T][RT][33]
  const int tid = threadIdx.x;
  const int ncols = swap ? I : O, nrows = swap ? O : I;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * RT;
  const size_t sstride = (size_t)K * N;
  const int nload = T * RT * 32;
  for (int e = tid; e < nload; e += 256) {
    const int c_l = e & 31;
    const int r_l = (e >> 5) % RT;
    const int t = (e >> 5) / RT;
    const int c = c0 + c_l, r = r0 + r_l;
    float v = 0.f;
    if (c < ncols && r < nrows) {
      const float* sp = slab + ((size_t)t * Cg + r) * N + c;
      for (int s = 0; s < S; ++s) v += sp[s * sstride];
    }
    tile[(t * RT + r_l) * 33 + c_l] = v;
  }
  __syncthreads();
  const int nout = 32 * RT * Tp;
  for (int w = tid; w < nout; w += 256) {
    const int tapo = w % Tp;
    const int rest = w / Tp;
    int o_l, i_l, r_l, c_l;
    if (swap) { i_l = rest & 31; o_l = rest >> 5; r_l = o_l; c_l = i_l; }
    else { i_l = rest % RT; o_l = rest / RT; r_l = i_l; c_l = o_l; }
    const int o = swap ? r0 + o_l : c0 + o_l;
    const int i = swap ? c0 + i_l : r0 + i_l;
    if (o >= O || i >= I) continue;
    float v;
    if (fold) {
      const int ky = tapo / 3, kx = tapo - ky * 3;
      v = 0.f;
#pragma unroll
      for (int ay = 0; ay < 2; ++ay)
#pragma unroll
        for (int ax = 0; ax < 2; ++ax) {
          const int t = ((2 - ky) + ay) * 4 + (2 - kx) + ax;  // ky=0:{2,3} ky=1:{1,2} ky=2:{0,1}
          v += tile[(t * RT + r_l) * 33 + c_l];
        }
    } else {
      v = tile[(tapo * RT + r_l) * 33 + c_l];
    }
    float* gp = grad + ((size_t)o * I_total + i_off + i) * Tp + tapo;
    *gp = accumulate ? *gp + v : v;
  }
}

// OIHW -> packed P[t][Ip][Op] through an LDS tile of 64 cout x 8 cin x taps.  Round 3: 64 instead of 32 output channels per
// tile (a packed row segment is 256 contiguous bytes), 16-byte accesses on both sides where the shape allows, the tap count a
// template parameter (no integer division by a runtime value per element).  D_NET256's re-pack: see profiles/README.md.
template <int TP>   // taps of the parameter (16, 9, or 0 = any: scalar accesses)
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, float* __restrict__ packed, int O, int I, int Tp_rt,
                                          int Ip, int Op, int T, int mode, int bx, int by, float* tile) {
  const int Tp = TP ? TP : Tp_rt;
  const int tid = threadIdx.x;
  const int o0 = bx * 64, i0 = by * 8;
  const int ostride = 8 * Tp + 1;
  if (TP && (I & 3) == 0 && (((size_t)w) & 15) == 0) {
    // a row of the tile is 8 * TP contiguous floats of w (2 * TP float4); i0 is a multiple of 8 and I of 4: 16-byte aligned
    constexpr int F4 = 2 * (TP ? TP : 1);
    for (int e = tid; e < 64 * F4; e += 256) {
      const int o_l = e / F4, f = e - o_l * F4;
      const int o = o0 + o_l;
      const int i_first = i0 + (f * 4) / Tp;            // cin of the first of the four floats
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o < O && i_first < I) {
        const size_t off = ((size_t)o * I + i0) * Tp + f * 4;
        if (((size_t)o * I + i0) * Tp + f * 4 + 3 < (size_t)(o + 1) * I * Tp) v = *reinterpret_cast<const f32x4*>(w + off);
        else
#pragma unroll
          for (int j = 0; j < 4; ++j) if (off + j < (size_t)(o + 1) * I * Tp) v[j] = w[off + j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[o_l * ostride + f * 4 + j] = v[j];
    }
  } else {
    const int nload = 64 * 8 * Tp;
    for (int e = tid; e < nload; e += 256) {
      const int tapo = e % Tp;
      const int i_l = (e / Tp) & 7;
      const int o_l = e / (Tp * 8);
      const int o = o0 + o_l, i = i0 + i_l;
      tile[o_l * ostride + i_l * Tp + tapo] = (o < O && i < I) ? w[((size_t)o * I + i) * Tp + tapo] : 0.f;
    }
  }
  __syncthreads();
  // (t, i) rows of 64 output channels = 16 float4
  const int nout = T * 8 * 16;
  for (int e = tid; e < nout; e += 256) {
    const int o4 = e & 15;
    const int i_l = (e >> 4) & 7;
    const int t = e >> 7;
    const int o = o0 + o4 * 4, i = i0 + i_l;
    if (o >= Op || i >= Ip) continue;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* tp = tile + (o4 * 4 + j) * ostride + i_l * Tp;
      float x = 0.f;
      if (mode == S2I_PACK_UPFOLD) {
        // effective tap k4 sums parameter taps: 0:{2} 1:{1,2} 2:{0,1} 3:{0}
        const int k4y = t >> 2, k4x = t & 3;
        const int ylo = k4y == 0 ? 2 : (k4y == 1 ? 1 : 0), yhi = k4y == 0 ? 2 : (k4y == 1 ? 2 : (k4y == 2 ? 1 : 0));
        const int xlo = k4x == 0 ? 2 : (k4x == 1 ? 1 : 0), xhi = k4x == 0 ? 2 : (k4x == 1 ? 2 : (k4x == 2 ? 1 : 0));
        for (int ky = ylo; ky <= yhi; ++ky)
          for (int kx = xlo; kx <= xhi; ++kx) x += tp[ky * 3 + kx];
      } else {
        x = tp[t];
      }
      v[j] = x;
    }
    *reinterpret_cast<f32x4*>(packed + ((size_t)t * Ip + i) * Op + o) = v;   // Op is a multiple of 4 and so is o
  }
}

__device__ __forceinline__ void pack_tile_any(const float* __restrict__ w, float* __restrict__ packed, int O, int I, int Tp,
                                              int Ip, int Op, int T, int mode, int bx, int by, float* tile) {
  if (Tp == 16) pack_tile<16>(w, packed, O, I, Tp, Ip, Op, T, mode, bx, by, tile);
  else if (Tp == 9) pack_tile<9>(w, packed, O, I, Tp, Ip, Op, T, mode, bx, by, tile);
  else pack_tile<0>(w, packed, O, I, Tp, Ip, Op, T, mode, bx, by, tile);
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                                          int O, int I, int Tp, int Ip, int Op, int T, int mode) {
  extern __shared__ float tile[];  // [64][8*Tp + 1]
  pack_tile_any(w, packed, O, I, Tp, Ip, Op, T, mode, blockIdx.x, blockIdx.y, tile);
}

// every conv weight of one network in ONE launch (after the fused Adam step): the table lives in device memory; item.gx
// counts 64-channel tiles
__global__ __launch_bounds__(256) void pack_weight_batched_kernel(const s2i_pack_item* __restrict__ items, int n) {
  extern __shared__ float tile[];
  int k = 0;
  while (k + 1 < n && (int)blockIdx.x >= items[k + 1].block0) ++k;  // n is a few dozen; block0 ascending
  const s2i_pack_item it = items[k];
  const int local = blockIdx.x - it.block0;
  const int T = it.mode == S2I_PACK_UPFOLD ? 16 : it.KH * it.KW;
  pack_tile_any(it.w, it.packed, it.O, it.I, it.KH * it.KW, it.Ip, (it.O + 3) & ~3, T, it.mode, local % it.gx, local / it.gx,
                tile);
}

// ---- host-side planning ------------------------------------------------------------------------
struct FwdPlan {
  int T, K, Ca, Ho, Wo, M, nphases, tile, bm, gridM, gridN, nchunks, splitk, cps;
  long long Mrows;
};

// K split of a launch of `blocks` output tiles: three 256-thread blocks fit per CU, so split K until about 768 blocks exist
static int fwd_splitk(long long blocks, int nchunks, int nosplit, int min_cps) {
  int splitk = 1;
  if (blocks < 512 && nchunks >= 16 && !nosplit) {
    splitk = (int)(768 / blocks);
    if (splitk > nchunks / min_cps) splitk = nchunks / min_cps;
    if (splitk > 64) splitk = 64;
    if (splitk < 1) splitk = 1;
  }
  return splitk;
}

int plan_fwd(const s2i_conv_desc* d, FwdPlan* pl) {
  S2I_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->N > 0, "conv: non-positive extent");
  S2I_REQUIRE(d->Cx >= 0 && d->Cc >= 0 && (d->Cx % 4) == 0 && (d->Cc % 4) == 0 && d->Cx + d->Cc > 0,
              "conv: channel counts must be multiples of 4 (Cx=%d Cc=%d)", d->Cx, d->Cc);
  S2I_REQUIRE(s2i_is_pow2(d->H) && s2i_is_pow2(d->W), "conv: spatial extents must be powers of two");
  pl->Ca = d->Cx + d->Cc;
  pl->nphases = 1;
  switch (d->kind) {
    case S2I_CONV_K1: pl->T = 1; pl->Ho = d->H; pl->Wo = d->W; break;
    case S2I_CONV_K3S1: pl->T = 9; pl->Ho = d->H; pl->Wo = d->W; break;
    case S2I_CONV_K4S2:
      S2I_REQUIRE(d->H >= 2 && d->W >= 2, "conv k4s2: extent < 2");
      pl->T = 16; pl->Ho = d->H / 2; pl->Wo = d->W / 2; break;
    case S2I_TCONV_K4S2: pl->T = 4; pl->Ho = d->H; pl->Wo = d->W; pl->nphases = 4; break;
    case S2I_CONV_1D:
      S2I_REQUIRE(d->kw >= 1 && d->kw <= 31 && d->stride >= 1 && d->pad >= 0, "conv1d: bad kw/stride/pad");
      S2I_REQUIRE(d->wmode == 0 && !d->flip, "conv1d: forward only");
      pl->T = d->kw; pl->Ho = d->H; pl->Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
      S2I_REQUIRE(pl->Wo >= 1 && s2i_is_pow2(pl->Wo), "conv1d: output width %d is not a power of two", pl->Wo);
      break;
    default: S2I_FAIL("conv: unknown kind %d", d->kind);
  }
  const long long M = (long long)d->B * pl->Ho * pl->Wo;
  S2I_REQUIRE(M * 4 < (1ll << 31), "conv: too many rows");
  pl->M = (int)M;
  pl->Mrows = M * pl->nphases;
  pl->K = pl->T * pl->Ca;
  if (d->wmode == 0) {
    S2I_REQUIRE(d->wR >= pl->Ca, "conv: wR (%d) smaller than the gathered channels (%d)", d->wR, pl->Ca);
    S2I_REQUIRE(d->ldw >= d->N && d->ldw % 4 == 0, "conv: ldw %d too small for N %d", d->ldw, d->N);
  } else {
    S2I_REQUIRE(d->wR >= d->N, "conv(T): wR (%d) < N (%d)", d->wR, d->N);
    S2I_REQUIRE(d->ldw >= pl->Ca, "conv(T): ldw (%d) < gathered channels (%d)", d->ldw, pl->Ca);
  }
  S2I_REQUIRE(d->ldy >= d->N, "conv: ldy < N");
  S2I_REQUIRE(!(d->stats && (d->act != S2I_ACT_NONE)), "conv: stats epilogue needs act NONE");
  if (d->stats && d->groups > 1) {
    // independent BatchNorm batches stacked along the rows: a row tile must not straddle two of them (tile heights are
    // checked per candidate below; 96-row tiles exist only for N > 64)
    S2I_REQUIRE(d->kind != S2I_TCONV_K4S2 && (M % d->groups) == 0 &&
                    (((M / d->groups) % 128) == 0 || (d->N > 64 && ((M / d->groups) % 96) == 0 && d->tile_rows != 128)),
                "conv: %lld rows do not split into %d BatchNorm groups of whole row tiles", M, d->groups);
    S2I_REQUIRE((d->N % 4) == 0, "conv: grouped statistics need N %% 4 == 0");
  }
  pl->tile = d->N > 64 ? 0 : (d->N > 32 ? 1 : 2);
  const int BN = pl->tile == 0 ? 128 : (pl->tile == 1 ? 64 : 32);
  pl->gridN = s2i_cdiv(d->N, BN);
  pl->nchunks = s2i_cdiv(pl->K, 32);
  // Rows per tile.  The chip holds 768 blocks at a time (three per CU); a launch whose tiles x K-splits fill whole
  // rounds of them runs at 121 - 125 TFLOP/s, one that ends on 0.5 or 0.75 of a round at ~100
  // (profiles/r02_f32_per_launch_table.txt).  The discriminators' stacked passes have 72 = 8 x 9 images, so 128-row
  // tiles give 9 x 2^k of them (576, 1152: 0.75 / 1.5 rounds) where 96-row tiles give 3 x 2^k (768, 1536).  (192 x 128
  // tiles need 168+ registers: no third block per CU, and at two per CU a round holds the same rows as with 128.)
  // Candidates are priced as rounds x (chunks per block + a fixed prologue / epilogue share) x rows, the smaller
  // tile with the measured relative cost of its matrix loop.
  const int forced_bm = d->tile_rows ? d->tile_rows : s2i_tune(S2I_TUNE_FWD_BM, 0);
  S2I_REQUIRE(forced_bm == 0 || forced_bm == 96 || forced_bm == 128, "conv: tile_rows must be 0, 96 or 128");
  const int min_cps = s2i_tune(S2I_TUNE_FWD_MIN_CPS, 4);
  static const int cand_bm[2] = {128, 96};
  double best = 1e300;
  int best_bm = 128, best_split = 1;
  for (int c = 0; c < 2; ++c) {
    const int bm = cand_bm[c];
    if (forced_bm ? bm != forced_bm : false) continue;
    if (bm == 96 && pl->tile != 0) continue;                       // 96 x 128 only (four waves side by side)
    if (bm != 128 && (d->kind == S2I_CONV_1D || M < 2 * bm)) continue;
    if (d->stats && d->groups > 1 && ((M / d->groups) % bm) != 0) continue;
    const long long blocks = (long long)s2i_cdiv(M, bm) * pl->gridN * pl->nphases;
    const int sk0 = fwd_splitk(blocks, pl->nchunks, d->nosplit, min_cps);
    const int cps = s2i_cdiv(pl->nchunks, sk0), sk = s2i_cdiv(pl->nchunks, cps);
    const double rounds = (double)((blocks * sk + 767) / 768);
    const double rel = bm == 128 ? 1.0 : 0.80;   // time of one chunk of a block, 128 rows = 1
    // slab write + read at ~4 TB/s in units of one chunk round of the chip (768 x 128 x 128 x 32 MACs at 122 TFLOP/s = 6.6 us)
    const double slab = sk > 1 ? 3.0e-7 * sk * (double)pl->Mrows * d->N : 0.0;
    const double cost = rounds * (cps + 3.0) * rel + slab + (sk > 1 ? 2.0 : 0.0);
    if (cost < best * (bm == 128 ? 1.0 : 0.97)) { best = cost; best_bm = bm; best_split = sk; }
  }
  S2I_REQUIRE(best < 1e300 || !(d->stats && d->groups > 1 && ((M / d->groups) % 128) != 0),
              "conv: no tile height fits the %d BatchNorm groups of %lld rows", d->groups, M / (d->groups > 0 ? d->groups : 1));
  if (best == 1e300) { best_bm = 128; best_split = fwd_splitk((long long)s2i_cdiv(M, 128) * pl->gridN * pl->nphases, pl->nchunks, d->nosplit, min_cps); }
  pl->bm = best_bm;
  pl->gridM = s2i_cdiv(M, pl->bm);
  const int splitk = best_split;
  pl->cps = s2i_cdiv(pl->nchunks, splitk);
  pl->splitk = s2i_cdiv(pl->nchunks, pl->cps);
  return 0;
}

struct WgPlan {
  int T, K, Cin, Ho, Wo, M, tile, gridK, gridN, nchunks, splitk, cps, small_n, rows3, bn3;
};

// planes: 0 fp32 operands, 1..3 split-bf16 products (or one operand bf16), 16 both operands stored as bf16
int plan_wgrad(const s2i_wgrad_desc* d, WgPlan* pl, int planes = 0) {
  S2I_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->N > 0, "wgrad: non-positive extent");
  S2I_REQUIRE((d->Ca % 4) == 0 && (d->Cc % 4) == 0 && (d->N % 4) == 0 && d->Ca + d->Cc > 0,
              "wgrad: channel counts must be multiples of 4 (Ca=%d Cc=%d N=%d)", d->Ca, d->Cc, d->N);
  S2I_REQUIRE(s2i_is_pow2(d->H) && s2i_is_pow2(d->W), "wgrad: spatial extents must be powers of two");
  pl->Cin = d->Ca + d->Cc;
  switch (d->kind) {
    case S2I_CONV_K1: pl->T = 1; pl->Ho = d->H; pl->Wo = d->W; break;
    case S2I_CONV_K3S1: pl->T = 9; pl->Ho = d->H; pl->Wo = d->W; break;
    case S2I_CONV_K4S2: pl->T = 16; pl->Ho = d->H / 2; pl->Wo = d->W / 2; break;
    default: S2I_FAIL("wgrad: unsupported gather kind %d", d->kind);
  }
  const long long M = (long long)d->B * pl->Ho * pl->Wo;
  S2I_REQUIRE(M < (1ll << 30), "wgrad: too many rows");
  pl->M = (int)M;
  pl->K = pl->T * pl->Cin;
  S2I_REQUIRE(d->ldg >= d->N, "wgrad: ldg < N");
  // consistency of the OIHW target with the GEMM result
  const int taps_param = d->KH * d->KW;
  if (d->fold) S2I_REQUIRE(d->KH == 3 && d->KW == 3 && pl->T == 16, "wgrad: fold needs 3x3 param / 4x4 taps");
  else S2I_REQUIRE(taps_param == pl->T, "wgrad: taps mismatch (%d vs %d)", taps_param, pl->T);
  if (d->swap) S2I_REQUIRE(pl->Cin >= d->O && d->N == d->I, "wgrad(swap): shape mismatch");
  else S2I_REQUIRE(pl->Cin >= d->I && d->N >= d->O, "wgrad: shape mismatch");
  pl->tile = d->N > 64 ? 0 : (d->N > 32 ? 1 : 2);
  if (pl->K <= 64 && d->N > 32 && d->N <= 64) pl->tile = 3;  // first discriminator conv: 16 taps x (3+1) channels
  // K = 288 (3x3 taps x 32 channels, the generator's last stage) wastes a quarter of three 128-row tiles: 96-row tiles
  if (!planes && pl->K % 96 == 0 && d->N <= 64 && s2i_cdiv(pl->K, 128) * 128 * 5 > pl->K * 6) pl->tile = d->N > 32 ? 4 : 5;
  const int BN = pl->tile == 0 ? 128 : ((pl->tile == 2 || pl->tile == 5) ? 32 : 64);
  const int BM = pl->tile == 3 ? 64 : (pl->tile >= 4 ? 96 : 128);
  pl->gridK = s2i_cdiv(pl->K, BM);
  pl->gridN = s2i_cdiv(d->N, BN);
  pl->nchunks = s2i_cdiv(M, 32);
  const long long tiles = (long long)pl->gridK * pl->gridN;
  // three resident blocks per CU hide each other's load latency: split the pixel range so that tiles x splits fill whole
  // rounds of the chip's 768 block slots (512 tiles x 1 = 0.67 of a round ran at 108 TFLOP/s, profiles/r02_f32_per_launch_table.txt),
  // priced as rounds x (chunks per block + a fixed share) + the fp32 slabs each split writes and the finish pass reads
  // (units: one chunk round of the chip, 6.6 us)
  int splitk = 1;
  {
    int smax = pl->nchunks / 4;
    if (smax > 256) smax = 256;
    if (smax < 1) smax = 1;
    double best = 1e300;
    for (int sc = 1; sc <= smax; ++sc) {
      const int cps = s2i_cdiv(pl->nchunks, sc), se = s2i_cdiv(pl->nchunks, cps);
      if (se != sc) continue;                       // same effective split as a smaller candidate
      const double rounds = (double)((tiles * se + 767) / 768);
      const double cost = rounds * (cps + 3.0) + 3.0e-7 * se * (double)pl->K * d->N;
      if (cost < best) { best = cost; splitk = se; }
      if (tiles * sc > 4 * 768) break;
    }
  }
  if (!planes && pl->tile == 0 && (pl->K % 256) == 0 && !d->a_act && s2i_tune(S2I_TUNE_WGRAD_BM, 0) != 128) {
    // fp32 256 x 128 tiles on 512-thread blocks (two per CU, 512 slots): half the `g` re-reads of the 128 x 128 form; a round
    // of chunks takes 1.26x as long for 1.33x the work (8.3 us against 6.6: tools/wgrad_bench.py, 103 -> 119 TFLOP/s on
    // D_NET256's first stacked weight gradient).  Taken where the same cost model prices it lower.
    const long long t2 = (long long)(pl->K / 256) * pl->gridN;
    int smax = pl->nchunks / 4, s8 = 1;
    if (smax > 256) smax = 256;
    if (smax < 1) smax = 1;
    double best8 = 1e300, cost0 = 1e300;
    for (int sc = 1; sc <= smax; ++sc) {
      const int cps = s2i_cdiv(pl->nchunks, sc), se = s2i_cdiv(pl->nchunks, cps);
      if (se != sc) continue;
      const double cost = (double)((t2 * se + 511) / 512) * (cps + 3.0) * 1.26 + 3.0e-7 * se * (double)pl->K * d->N;
      if (cost < best8) { best8 = cost; s8 = se; }
      if (t2 * sc > 4 * 512) break;
    }
    {
      const int cps = s2i_cdiv(pl->nchunks, splitk), se = s2i_cdiv(pl->nchunks, cps);
      cost0 = (double)((tiles * se + 767) / 768) * (cps + 3.0) + 3.0e-7 * se * (double)pl->K * d->N;
    }
    // 256 x 256 tiles on one 1024-thread block per CU (256 slots) where 256 divides N: a round of chunks takes 1.22x the
    // 128 x 128 round for 1.33x the work (D_NET256's three middle layers 0.650 -> 0.628 ms, profiles/r03_f32_wgrad_tile_heights.txt)
    int s9 = 1;
    double best9 = 1e300;
    if ((d->N % 256) == 0) {
      const long long t3 = (long long)(pl->K / 256) * (d->N / 256);
      for (int sc = 1; sc <= smax; ++sc) {
        const int cps = s2i_cdiv(pl->nchunks, sc), se = s2i_cdiv(pl->nchunks, cps);
        if (se != sc) continue;
        const double cost = (double)((t3 * se + 255) / 256) * (cps + 3.0) * 1.22 + 3.0e-7 * se * (double)pl->K * d->N;
        if (cost < best9) { best9 = cost; s9 = se; }
        if (t3 * sc > 4 * 256) break;
      }
    }
    const int force = s2i_tune(S2I_TUNE_WGRAD_BM, 0);   // 0 model, 128 / 256 / 512 (= 256 x 256) forced where eligible
    if ((force == 512 && best9 < 1e300) || (force == 0 && best9 < best8 && best9 < cost0)) {
      pl->tile = 9;
      pl->gridK = pl->K / 256;
      pl->gridN = d->N / 256;
      splitk = s9;
    } else if (best8 < cost0 || force == 256 || force == 512) {
      pl->tile = 8;
      pl->gridK = pl->K / 256;
      splitk = s8;
    }
  }
  pl->cps = s2i_cdiv(pl->nchunks, splitk);
  pl->splitk = s2i_cdiv(pl->nchunks, pl->cps);
  if (planes == 16 && pl->tile == 0 && (pl->K % 256) == 0 && (pl->Cin % 8) == 0 && d->Cc == 0 && (d->N % 8) == 0 &&
      (d->ldg % 8) == 0 && s2i_tune(S2I_TUNE_WGRAD16_BM, 0) != 128) {
    // both operands bf16: 256 x 128 tiles on 512-thread blocks, two per CU (512 slots), 64-pixel stages -- where that is
    // cheaper than the 128 x 128 plan above under one model for both (us; measured on the config-4 layers,
    // tools/wgrad16_bench.py: a round of 64-pixel stages takes ~2.5 us with 768 blocks of 128 x 128 and ~2.7 us with 512
    // blocks of 256 x 128; a block's prologue / epilogue is worth 3 resp. 2 stages; each split writes and re-reads a slab)
    const int nch64 = s2i_cdiv(M, 64);
    const double slab_us = 2.0e-6 * (double)pl->K * d->N;
    const int cps128 = s2i_cdiv(nch64, pl->splitk), se128 = s2i_cdiv(nch64, cps128);
    const double cost128 = (double)((tiles * se128 + 767) / 768) * (cps128 + 3.0) * 2.5 + slab_us * se128;
    const long long t2 = (long long)(pl->K / 256) * pl->gridN;
    int smax = nch64 / 4, best_s = 1;
    if (smax > 256) smax = 256;
    if (smax < 1) smax = 1;
    double best = 1e300;
    for (int sc = 1; sc <= smax; ++sc) {
      const int cps = s2i_cdiv(nch64, sc), se = s2i_cdiv(nch64, cps);
      if (se != sc) continue;
      const double cost = (double)((t2 * se + 511) / 512) * (cps + 2.0) * 2.7 + slab_us * se;
      if (cost < best) { best = cost; best_s = se; }
      if (t2 * sc > 4 * 512) break;
    }
    // 256 x 256 tiles, one 1024-thread block per CU (256 slots), where 256 divides N: ~2.15 us per round of stages
    int best3_s = 1;
    double best3 = 1e300;
    if ((d->N % 256) == 0) {
      const long long t3 = (long long)(pl->K / 256) * (d->N / 256);
      for (int sc = 1; sc <= smax; ++sc) {
        const int cps = s2i_cdiv(nch64, sc), se = s2i_cdiv(nch64, cps);
        if (se != sc) continue;
        const double cost = (double)((t3 * se + 255) / 256) * (cps + 3.0) * 2.15 + slab_us * se;
        if (cost < best3) { best3 = cost; best3_s = se; }
        if (t3 * sc > 4 * 256) break;
      }
    }
    const int force = s2i_tune(S2I_TUNE_WGRAD16_BM, 0);   // 0 model, 128 / 256 / 512 (= 256 x 256) forced where eligible
    int pick = 0;
    if (force == 512 && best3 < 1e300) pick = 7;
    else if (force == 256 || force == 512) pick = 6;
    else if (force == 0) {
      const double m = best3 < best ? best3 : best;
      if (m < cost128) pick = best3 < best ? 7 : 6;
    }
    if (pick) {
      const int bs = pick == 7 ? best3_s : best_s;
      pl->tile = pick;
      pl->gridK = pl->K / 256;
      if (pick == 7) pl->gridN = d->N / 256;
      // kept in 32-pixel chunks like the other plans (the launcher re-derives the 64-pixel stages from splitk)
      pl->cps = s2i_cdiv(pl->nchunks, bs);
      pl->splitk = s2i_cdiv(pl->nchunks, pl->cps);
    }
  }
  // thin 3x3 layers over wide maps: one kernel row per block, taps read from a staged row segment
  pl->rows3 = !planes && d->kind == S2I_CONV_K3S1 && d->Cc == 0 && (d->Ca == 32 || d->Ca == 64) && d->W >= 32 &&
              (d->N % 32) == 0 && d->N <= 128;
  if (pl->rows3) {
    pl->bn3 = (d->Ca == 64 && d->N > 64) ? 128 : (d->N > 32 ? 64 : 32);
    pl->gridN = s2i_cdiv(d->N, pl->bn3);
    // measured (24x128x128, 32->64): 512 / 768 / 1024 / 1536 blocks = 211 / 202 / 183 / 234 us; four 192-thread blocks or
    // three 256-thread blocks fill a CU, more only adds slab traffic
    const int target = d->Ca == 32 ? 1024 : 768;
    int sk = target / ((d->Ca == 32 ? 1 : 3) * pl->gridN);
    if (sk > pl->nchunks / 4) sk = pl->nchunks / 4;
    if (sk > 2048) sk = 2048;
    if (sk < 1) sk = 1;
    pl->cps = s2i_cdiv(pl->nchunks, sk);
    pl->splitk = s2i_cdiv(pl->nchunks, pl->cps);
  }
  // <= 4 output channels of a 3x3 conv: streamed on the vector units, one slab per block
  pl->small_n = d->kind == S2I_CONV_K3S1 && d->Cc == 0 && d->N == 4 && d->ldg == 4 &&
                (d->Ca == 16 || d->Ca == 32 || d->Ca == 64) && d->W >= 16 && M >= (1 << 15);
  if (pl->small_n) pl->splitk = 512;
  return 0;
}

template <int BM, int BN, int WM, int WN>
void launch_fwd(const IgemmP& p, dim3 grid, bool wt, bool ca32, hipStream_t st) {
  if (p.in_coef) {   // apply-on-load (conv_forward_impl admits it for !wt && ca32 only)
    hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, false, true, true>), grid, dim3(256), 0, st, p);
    return;
  }
  if (wt) {
    if (ca32) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, true, false>), grid, dim3(256), 0, st, p);
  } else {
    if (ca32) hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, false, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, WM, WN, false, false>), grid, dim3(256), 0, st, p);
  }
}

}  // namespace

// thin layers (one lane per output pixel, weights as a scalar table in the workspace): 1 = few outputs, 2 = 4 inputs
static int thin_kind(const s2i_conv_desc* d, const FwdPlan& pl) {
  if (d->Cc != 0 || d->stats || pl.M < 4096 || pl.splitk != 1) return 0;
  // measured (bf16 mode, batch 48): few outputs from 16 / 32 channels 113 / 66 us against 273 / 135 us for the
  // lanes-per-pixel kernel; from 64 channels the lane-per-pixel reads (128-byte pixels, one 16-byte piece per instruction)
  // thrash L1: 703 against 280 us, so that case stays with small_n_conv_kernel.  4 inputs to 16 / 32 outputs 78 / 43 us
  // against 225 / 64 us on the matrix kernel; to 64 outputs (first discriminator conv, 4096 FMAs per pixel) the vector
  // units tie with the fp32 matrix kernel (310 vs 315 us) and lose at batch 48 (152 vs 113): not taken.
  if (d->N <= 4 && (d->kind == S2I_CONV_K3S1 || d->kind == S2I_TCONV_K4S2) && (pl.Ca % 8) == 0 && pl.Ca <= 32) return 1;
  if (pl.Ca == 4 && d->kind == S2I_CONV_K3S1 && (d->N == 16 || d->N == 32) && (d->ldy % 8) == 0) return 2;
  return 0;
}
// transposed conv to <= 4 channels from 64 stored channels on maps of whole 8 x 8 tiles: tconv_n4_tile_kernel
static bool tile_n4_ok(const s2i_conv_desc* d, const FwdPlan& pl) {
  return d->kind == S2I_TCONV_K4S2 && d->N <= 4 && d->Cc == 0 && !d->stats && pl.Ca == 64 && (d->H % 8) == 0 && (d->W % 8) == 0 &&
         pl.M >= 4096 && pl.splitk == 1;
}
// conv3x3 to <= 4 channels from 16 / 32 / 64 stored channels on maps of whole 16 x 16 tiles: conv3_n4_tile_kernel
static bool tile3_n4_ok(const s2i_conv_desc* d, const FwdPlan& pl) {
  return d->kind == S2I_CONV_K3S1 && d->N <= 4 && d->Cc == 0 && !d->stats && (pl.Ca == 16 || pl.Ca == 32 || pl.Ca == 64) &&
         (d->H % 16) == 0 && (d->W % 16) == 0 && pl.M >= 4096 && pl.splitk == 1;
}
static size_t thin_table_floats(const s2i_conv_desc* d, const FwdPlan& pl, int tk) {
  return tk == 1 ? (size_t)pl.nphases * pl.T * pl.Ca * 4 : (size_t)pl.T * 4 * d->N;
}

// bf16-mode image layers on the matrix cores with pixels as columns: 1 = few outputs from bf16 input, 2 = fp32 NHWC4 input to
// bf16 output (the dtype combination decides: these are the edges of the bf16 activation mode only)
static int rgb_kind(const s2i_conv_desc* d, const FwdPlan& pl, int x16, int y16) {
  if (d->Cc != 0 || d->stats || pl.M < 4096) return 0;
  if (x16 && !y16 && d->N <= 4 && (d->kind == S2I_CONV_K3S1 || d->kind == S2I_TCONV_K4S2) &&
      (pl.Ca == 16 || pl.Ca == 32 || pl.Ca == 64))
    return 1;
  if (!x16 && y16 && pl.Ca == 4 && (d->kind == S2I_CONV_K3S1 || d->kind == S2I_CONV_K4S2) &&
      (d->N == 16 || d->N == 32 || d->N == 64) && (d->ldy % 4) == 0)
    return 2;
  return 0;
}
static size_t rgb_afrag_elems(const s2i_conv_desc* d, const FwdPlan& pl, int rk) {
  if (rk == 1) return (size_t)pl.nphases * (pl.T * pl.Ca / 16) * 64 * 8;
  return (size_t)((d->N + 31) / 32) * (d->kind == S2I_CONV_K4S2 ? 4 : 3) * 64 * 8;
}

extern "C" size_t s2i_conv_workspace_bytes(const s2i_conv_desc* d) {
  FwdPlan pl;
  if (plan_fwd(d, &pl)) return 0;
  const int tk = thin_kind(d, pl);
  // the dtype-dependent rgb kernels need at most this much as well (bf16 fragments; sized for either)
  size_t rgb = 0;
  if (rgb_kind(d, pl, 1, 0)) rgb = rgb_afrag_elems(d, pl, 1) * 2;
  if (rgb_kind(d, pl, 0, 1)) rgb = rgb_afrag_elems(d, pl, 2) * 2;
  // the caller does not know which kernel the dtypes will select: the largest requirement of the candidates
  size_t need = pl.splitk > 1 ? (size_t)pl.splitk * pl.Mrows * d->N * sizeof(float) : 0;
  if (tk) { const size_t tb = thin_table_floats(d, pl, tk) * sizeof(float); need = tb > need ? tb : need; }
  if (tile_n4_ok(d, pl)) { const size_t tb = (size_t)4 * 4 * 64 * 4 * sizeof(float); need = tb > need ? tb : need; }
  if (tile3_n4_ok(d, pl)) { const size_t tb = (size_t)9 * pl.Ca * 4 * sizeof(float); need = tb > need ? tb : need; }
  if (rgb || tk || tile_n4_ok(d, pl) || tile3_n4_ok(d, pl)) return rgb > need ? rgb : need;
  return pl.splitk > 1 ? (size_t)pl.splitk * pl.Mrows * d->N * sizeof(float) : 0;
}

// number of rows the stats pass writes; split-K layers take the column-stats kernel instead
static int stat_parts_for(const FwdPlan& pl, int groups) {
  if (groups < 1) groups = 1;
  if (pl.splitk > 1) {
    int ppg = s2i_cdiv(pl.Mrows / groups, 8);  // split-K layers have few rows: keep the reduce+stats pass wide
    if (ppg > 512 / groups) ppg = 512 / groups;
    if (ppg < 1) ppg = 1;
    return ppg * groups;
  }
  return pl.gridM * pl.nphases;
}

extern "C" int s2i_conv_stat_parts(const s2i_conv_desc* d) {
  FwdPlan pl;
  if (plan_fwd(d, &pl)) return -1;
  return stat_parts_for(pl, d->groups);
}

extern "C" int s2i_conv_forward(const s2i_conv_desc* d, const float* x, const float* cvec, const float* w,
                                const float* bias, float* y, float* part, void* ws, size_t ws_bytes,
                                void* stream) {
  return s2i_conv_forward_cls(d, x, cvec, w, bias, nullptr, y, part, ws, ws_bytes, stream);
}

static int conv_forward_impl(const s2i_conv_desc* d, const float* x, const float* cvec, const float* w,
                             const unsigned short* wsp, int planes, int np, int kp, const float* bias,
                             const float* cls_bias, float* y, float* part, void* ws, size_t ws_bytes, void* stream,
                             int x16 = 0, int y16 = 0, const float* in_coef = nullptr);

extern "C" int s2i_conv_forward_cls(const s2i_conv_desc* d, const float* x, const float* cvec, const float* w,
                                    const float* bias, const float* cls_bias, float* y, float* part, void* ws,
                                    size_t ws_bytes, void* stream) {
  S2I_REQUIRE(w != nullptr, "conv: null weight");
  return conv_forward_impl(d, x, cvec, w, nullptr, 0, 0, 0, bias, cls_bias, y, part, ws, ws_bytes, stream);
}

extern "C" int s2i_conv_forward_in(const s2i_conv_desc* d, const float* x_raw, const float* in_coef, const float* w, float* y,
                                   float* part, void* ws, size_t ws_bytes, void* stream) {
  S2I_REQUIRE(w != nullptr && in_coef != nullptr, "conv(apply-on-load): null weight / coefficient table");
  return conv_forward_impl(d, x_raw, nullptr, w, nullptr, 0, 0, 0, nullptr, nullptr, y, part, ws, ws_bytes, stream, 0, 0, in_coef);
}

extern "C" int s2i_conv_split_eligible(const s2i_conv_desc* d) {
  FwdPlan pl;
  if (plan_fwd(d, &pl)) return 0;
  const bool small_n = d->N <= 4 && d->Cc == 0 && !d->stats && (d->kind == S2I_CONV_K3S1 || d->kind == S2I_TCONV_K4S2) &&
                       (pl.Ca == 16 || pl.Ca == 32 || pl.Ca == 64) && pl.M >= 4096;
  return (pl.Ca % 32) == 0 && (d->Cc % 32) == 0 && !small_n;
}

extern "C" int s2i_conv_forward_split(const s2i_conv_desc* d, const float* x, const float* cvec,
                                      const unsigned short* wsplit, int planes, int np, int kp, const float* bias,
                                      const float* cls_bias, float* y, float* part, void* ws, size_t ws_bytes,
                                      void* stream) {
  S2I_REQUIRE(wsplit != nullptr && planes >= 1 && planes <= 3, "conv(split): need 1 to 3 bf16 planes");
  S2I_REQUIRE(np > 0 && kp > 0 && (kp % 8) == 0, "conv(split): weight rows must be multiples of 8 bf16 (kp=%d)", kp);
  S2I_REQUIRE(s2i_conv_split_eligible(d), "conv(split): layer not eligible (gathered channels must be multiples of 32)");
  return conv_forward_impl(d, x, cvec, nullptr, wsplit, planes, np, kp, bias, cls_bias, y, part, ws, ws_bytes, stream);
}

extern "C" int s2i_split_packed_weight(const float* packed, int T, int R, int C, int planes, unsigned short* out_rc,
                                       unsigned short* out_cr, void* stream) {
  S2I_REQUIRE(packed && (out_rc || out_cr) && T > 0 && R > 0 && C > 0 && planes >= 1 && planes <= 3,
              "split_packed_weight: bad args");
  dim3 grid(s2i_cdiv(C, 32), s2i_cdiv(R, 32), T);
  hipLaunchKernelGGL(split_packed_kernel, grid, dim3(256), 0, (hipStream_t)stream, packed, out_rc, out_cr, R, C, planes,
                     (long long)T * R * C);
  S2I_LAUNCH_CHECK("split_packed_weight");
  return 0;
}

template <int BM, int BN, int WM, int WN>
static void launch_split(const IgemmP& p, dim3 grid, int planes, hipStream_t st) {
  if (planes == 1) hipLaunchKernelGGL((igemm_fwd_split_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), 0, st, p);
  else if (planes == 2) hipLaunchKernelGGL((igemm_fwd_split_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((igemm_fwd_split_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, st, p);
}

extern "C" int s2i_conv_forward_dt(const s2i_conv_desc* d, const void* x, int x_dtype, const float* cvec, const float* w,
                                   const float* bias, const float* cls_bias, void* y, int y_dtype, float* part, void* ws,
                                   size_t ws_bytes, void* stream) {
  S2I_REQUIRE(w != nullptr, "conv: null weight");
  S2I_REQUIRE((x_dtype == S2I_DT_F32 || x_dtype == S2I_DT_BF16) && (y_dtype == S2I_DT_F32 || y_dtype == S2I_DT_BF16),
              "conv: unknown dtype");
  return conv_forward_impl(d, (const float*)x, cvec, w, nullptr, 0, 0, 0, bias, cls_bias, (float*)y, part, ws, ws_bytes,
                           stream, x_dtype == S2I_DT_BF16, y_dtype == S2I_DT_BF16);
}

static int conv_forward_impl(const s2i_conv_desc* d, const float* x, const float* cvec, const float* w,
                             const unsigned short* wsp, int planes, int np, int kp, const float* bias,
                             const float* cls_bias, float* y, float* part, void* ws, size_t ws_bytes, void* stream,
                             int x16, int y16, const float* in_coef) {
  FwdPlan pl;
  if (plan_fwd(d, &pl)) return 1;
  S2I_REQUIRE(!cls_bias || (d->kind == S2I_CONV_K3S1 && pl.splitk == 1), "conv: class bias needs an unsplit 3x3 conv");
  S2I_REQUIRE(x != nullptr || d->Cx == 0, "conv: x is null");
  S2I_REQUIRE(d->Cc == 0 || cvec != nullptr, "conv: cvec is null but Cc > 0");
  S2I_REQUIRE((w || wsp) && y, "conv: null weight/output");
  S2I_REQUIRE(!d->stats || part, "conv: stats requested without a partial buffer");
  const size_t need = pl.splitk > 1 ? (size_t)pl.splitk * pl.Mrows * d->N * sizeof(float) : 0;
  S2I_REQUIRE(ws_bytes >= need && (need == 0 || ws), "conv: workspace too small (%zu < %zu)", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  IgemmP p;
  p.x = x; p.cvec = cvec; p.w = w; p.bias = bias; p.cls_bias = cls_bias; p.y = y; p.part = part; p.slab = (float*)ws;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cx = d->Cx; p.Cc = d->Cc; p.Ca = pl.Ca;
  p.Ho = pl.Ho; p.Wo = pl.Wo; p.lgWo = s2i_ilog2(pl.Wo); p.lgHoWo = s2i_ilog2(pl.Ho * pl.Wo);
  p.M = pl.M; p.N = d->N; p.K = pl.K; p.T = pl.T;
  p.kind = d->kind; p.flip = d->flip; p.act = d->act; p.stats = d->stats;
  p.splitk = pl.splitk; p.cps = pl.cps; p.nchunks = pl.nchunks;
  p.ldw = d->ldw; p.wR = d->wR; p.ldy = d->ldy; p.nparts = pl.gridM * pl.nphases;
  p.g_kw = d->kw; p.g_s = d->stride; p.g_pad = d->pad; p.wt = d->wmode != 0;
  p.Mrows = pl.Mrows;
  p.x16 = x16; p.y16 = y16;
  S2I_REQUIRE(!(wsp && (x16 || y16)), "conv(split): bf16 tensors go through s2i_conv_forward_bf16 / _dt");
  S2I_REQUIRE(!wsp || pl.bm == 128, "conv(split): the split-bf16 kernels have 128-row tiles (set tile_rows = 128)");
  p.wsp = wsp; p.wsp_np = np; p.wsp_kp = kp; p.wsp_plane = 0; p.wsp_bytes = 0;
  p.in_coef = in_coef; p.in_rows_per_group = pl.M;
  if (in_coef) {
    const int g = d->in_groups < 1 ? 1 : d->in_groups;
    S2I_REQUIRE(d->in_act == S2I_ACT_LRELU, "conv(apply-on-load): the producer's activation must be LeakyReLU (in_act=%d)", d->in_act);
    S2I_REQUIRE(!wsp && !x16 && !y16 && d->Cc == 0 && d->wmode == 0 && (pl.Ca % 32) == 0 && d->N > 4 && !bias && !cls_bias &&
                    d->kind != S2I_CONV_1D && d->kind != S2I_TCONV_K4S2,
                "conv(apply-on-load): fp32 forward of a stored tensor with 32 | Cx, no broadcast vector / bias");
    S2I_REQUIRE((pl.M % g) == 0 && (g == 1 || ((pl.M / g) % pl.bm) == 0),
                "conv(apply-on-load): %d rows do not split into %d producer groups of whole %d-row tiles", pl.M, g, pl.bm);
    p.in_rows_per_group = pl.M / g;
  }
  if (!wsp && !cls_bias && !in_coef && tile_n4_ok(d, pl) && !y16 && ws && ws_bytes >= (size_t)4 * 4 * 64 * 4 * sizeof(float)) {
    float* table = (float*)ws;
    const int total = 4 * 4 * 64 * 4;
    hipLaunchKernelGGL(thin_table_kernel, dim3(s2i_cdiv(total, 256)), dim3(256), 0, st, w, table, d->kind, d->flip, pl.T,
                       d->wmode != 0 ? 1 : 0, d->wR, d->ldw, 64, 4, 4);
    S2I_LAUNCH_CHECK("thin_table");
    hipLaunchKernelGGL((tconv_n4_tile_kernel<64>), dim3((d->H / 8) * (d->W / 8) * d->B), dim3(256), 0, st, p, (const float*)table);
    S2I_LAUNCH_CHECK("tconv_n4_tile");
    return 0;
  }
  if (!wsp && !cls_bias && !in_coef && tile3_n4_ok(d, pl) && !y16 && ws && ws_bytes >= (size_t)9 * pl.Ca * 4 * sizeof(float)) {
    float* table = (float*)ws;
    const int total = 9 * pl.Ca * 4;
    hipLaunchKernelGGL(thin_table_kernel, dim3(s2i_cdiv(total, 256)), dim3(256), 0, st, w, table, d->kind, d->flip, pl.T,
                       d->wmode != 0 ? 1 : 0, d->wR, d->ldw, pl.Ca, 4, 1);
    S2I_LAUNCH_CHECK("thin_table");
    const dim3 g3((d->H / 16) * (d->W / 16) * d->B);
    if (pl.Ca == 16) hipLaunchKernelGGL((conv3_n4_tile_kernel<16>), g3, dim3(256), 0, st, p, (const float*)table);
    else hipLaunchKernelGGL((conv3_n4_tile_kernel<32>), g3, dim3(256), 0, st, p, (const float*)table);
    S2I_LAUNCH_CHECK("conv3_n4_tile");
    return 0;
  }
  const int rk = (!wsp && !cls_bias && !in_coef) ? rgb_kind(d, pl, x16, y16) : 0;
  if (rk && ws && ws_bytes >= rgb_afrag_elems(d, pl, rk) * 2 && !(rk == 2 && bias)) {
    unsigned short* afrag = (unsigned short*)ws;
    const int total = (int)rgb_afrag_elems(d, pl, rk);
    const int KW = d->kind == S2I_CONV_K4S2 ? 4 : 3;
    const int MT = rk == 1 ? 1 : (d->N + 31) / 32, KS = rk == 1 ? pl.T * pl.Ca / 16 : (d->kind == S2I_CONV_K4S2 ? 4 : 3);
    hipLaunchKernelGGL(rgb_afrag_kernel, dim3(s2i_cdiv(total, 256)), dim3(256), 0, st, w, afrag, rk == 1 ? 0 : 1, d->kind, d->flip,
                       pl.T, d->wmode != 0 ? 1 : 0, d->wR, d->ldw, pl.Ca, KW, MT, KS, rk == 1 ? pl.nphases : 1, d->N);
    S2I_LAUNCH_CHECK("rgb_afrag");
    int blocks = s2i_cdiv(pl.M, 128);
    if (blocks > 2048) blocks = 2048;
    if (rk == 1) {
      dim3 g(blocks > 2048 / pl.nphases ? 2048 / pl.nphases : blocks, 1, pl.nphases);
      if (KS == 9) hipLaunchKernelGGL((rgb_out_kernel<9>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (KS == 18) hipLaunchKernelGGL((rgb_out_kernel<18>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (KS == 36) hipLaunchKernelGGL((rgb_out_kernel<36>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (KS == 4) hipLaunchKernelGGL((rgb_out_kernel<4>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (KS == 8) hipLaunchKernelGGL((rgb_out_kernel<8>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (KS == 16) hipLaunchKernelGGL((rgb_out_kernel<16>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else S2I_FAIL("rgb_out: unexpected k-step count %d", KS);
    } else {
      dim3 g(blocks);
      if (KS == 4 && MT == 2) hipLaunchKernelGGL((rgb_in_kernel<2, 4>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (KS == 4) hipLaunchKernelGGL((rgb_in_kernel<1, 4>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else if (MT == 2) hipLaunchKernelGGL((rgb_in_kernel<2, 3>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
      else hipLaunchKernelGGL((rgb_in_kernel<1, 3>), g, dim3(256), 0, st, p, (const unsigned short*)afrag);
    }
    S2I_LAUNCH_CHECK("rgb_conv");
    return 0;
  }
  const int tk = (!wsp && !cls_bias && !in_coef) ? thin_kind(d, pl) : 0;
  if (tk && ws && ws_bytes >= thin_table_floats(d, pl, tk) * sizeof(float)) {
    p.wt = d->wmode != 0;
    float* table = (float*)ws;
    const int Kk = tk == 1 ? pl.Ca : 4, Nn = tk == 1 ? 4 : d->N;
    const int total = (tk == 1 ? pl.nphases : 1) * pl.T * Kk * Nn;
    hipLaunchKernelGGL(thin_table_kernel, dim3(s2i_cdiv(total, 256)), dim3(256), 0, st, w, table, d->kind, d->flip, pl.T,
                       d->wmode != 0 ? 1 : 0, d->wR, d->ldw, Kk, Nn, tk == 1 ? pl.nphases : 1);
    S2I_LAUNCH_CHECK("thin_table");
    int blocks = s2i_cdiv(pl.M, 256);
    if (tk == 1) {
      if (blocks > 4096 / pl.nphases) blocks = 4096 / pl.nphases;
      // the table has 4 columns per (tap, channel); columns beyond N read the zero padding of the packed weights
      hipLaunchKernelGGL(thin_out_kernel, dim3(blocks, 1, pl.nphases), dim3(256), 0, st, p, (const float*)table);
    } else {
      if (blocks > 4096) blocks = 4096;
      if (d->N == 16) hipLaunchKernelGGL((thin_in_kernel<16>), dim3(blocks), dim3(256), 0, st, p, (const float*)table);
      else if (d->N == 32) hipLaunchKernelGGL((thin_in_kernel<32>), dim3(blocks), dim3(256), 0, st, p, (const float*)table);
      else hipLaunchKernelGGL((thin_in_kernel<64>), dim3(blocks), dim3(256), 0, st, p, (const float*)table);
    }
    S2I_LAUNCH_CHECK("thin_conv");
    return 0;
  }
  if (!wsp && d->N <= 4 && d->Cc == 0 && !d->stats && !cls_bias && (d->kind == S2I_CONV_K3S1 || d->kind == S2I_TCONV_K4S2) &&
      (pl.Ca == 16 || pl.Ca == 32 || pl.Ca == 64) && pl.M >= 4096) {
    // HBM-bound RGB-sized layers: VALU kernel instead of a 32-wide MFMA tile that is 7/8 padding
    p.wt = d->wmode != 0;
    const int lpp = pl.Ca / 4, ppw = 64 / lpp;
    int sblocks = s2i_cdiv(pl.M, 4 * ppw);
    if (sblocks > 2048 / pl.nphases) sblocks = 2048 / pl.nphases;   // 8 blocks per CU; the kernel strides over the rest
    dim3 sgrid(sblocks, 1, pl.nphases);
    const size_t shb = (size_t)pl.T * pl.Ca * 4 * sizeof(float);
    if (lpp == 4) hipLaunchKernelGGL((small_n_conv_kernel<4>), sgrid, dim3(256), shb, st, p);
    else if (lpp == 8) hipLaunchKernelGGL((small_n_conv_kernel<8>), sgrid, dim3(256), shb, st, p);
    else hipLaunchKernelGGL((small_n_conv_kernel<16>), sgrid, dim3(256), shb, st, p);
    S2I_LAUNCH_CHECK("small_n_conv");
    return 0;
  }
  dim3 grid(pl.gridM, pl.gridN, pl.nphases * pl.splitk);
  const bool wt = d->wmode != 0;
  const int wtaps = d->kind == S2I_TCONV_K4S2 ? 16 : pl.T;
  const unsigned long long xb = (unsigned long long)d->B * d->H * d->W * d->Cx * (x16 ? 2ull : 4ull);
  const unsigned long long wb = (unsigned long long)wtaps * d->wR * d->ldw * 4ull;
  S2I_REQUIRE(xb < 0x7ff00000ull && wb < 0x7ff00000ull, "conv: tensor exceeds the 2 GiB buffer-addressing window");
  p.x_bytes = (unsigned)xb;
  p.c_bytes = (unsigned)((unsigned long long)d->B * d->Cc * 4ull);
  p.w_bytes = (unsigned)wb;
  const bool ca32 = (pl.Ca % 32) == 0 && (d->Cc % 32) == 0;
  if (wsp) {
    const unsigned long long pe = (unsigned long long)wtaps * np * kp;  // elements per plane
    S2I_REQUIRE(pe * planes * 2ull < 0x7ff00000ull, "conv(split): weight planes exceed the buffer window");
    S2I_REQUIRE(kp >= pl.Ca && np >= d->N, "conv(split): weight planes %d x %d too small for N=%d K=%d", np, kp, d->N, pl.Ca);
    p.wsp_plane = (int)pe;
    p.wsp_bytes = (unsigned)(pe * planes * 2ull);
    if (pl.tile == 0) launch_split<128, 128, 2, 2>(p, grid, planes, st);
    else if (pl.tile == 1) launch_split<128, 64, 2, 2>(p, grid, planes, st);
    else launch_split<128, 32, 4, 1>(p, grid, planes, st);
    S2I_LAUNCH_CHECK("igemm_fwd_split");
  } else {
    if (pl.bm == 96) launch_fwd<96, 128, 1, 4>(p, grid, wt, ca32, st);
    else if (pl.tile == 0) launch_fwd<128, 128, 2, 2>(p, grid, wt, ca32, st);
    else if (pl.tile == 1) launch_fwd<128, 64, 2, 2>(p, grid, wt, ca32, st);
    else launch_fwd<128, 32, 4, 1>(p, grid, wt, ca32, st);
    S2I_LAUNCH_CHECK("igemm_fwd");
  }
  if (pl.splitk > 1) {
    if (d->stats && (d->N % 4) == 0 && (d->ldy % 4) == 0) {
      const int Q = d->N / 4;
      int cpb = 1;
      while (cpb < Q && cpb < 256) cpb <<= 1;
      const int groups = d->groups < 1 ? 1 : d->groups;
      const int nparts = stat_parts_for(pl, groups);
      hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3(nparts, (Q + cpb - 1) / cpb), dim3(256), 0, st,
                         (const float*)ws, pl.splitk, pl.Mrows, d->N, y, d->ldy, part, nparts, cpb, nparts / groups,
                         pl.Mrows / groups, y16);
      S2I_LAUNCH_CHECK("splitk_reduce_stats");
      return 0;
    }
    const long long total = pl.Mrows * d->N;
    int blocks = s2i_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, pl.splitk,
                       pl.Mrows, d->N, bias, d->act, y, d->ldy, y16);
    S2I_LAUNCH_CHECK("splitk_reduce");
    S2I_REQUIRE(!(d->stats && y16), "conv: bf16 output with statistics needs N %% 4 == 0 on a split-K layer");
    if (d->stats) return s2i_colstats(y, pl.Mrows, d->N, d->ldy, part, stat_parts_for(pl, 1), stream);
  }
  return 0;
}

extern "C" size_t s2i_wgrad_workspace_bytes(const s2i_wgrad_desc* d) {
  WgPlan pl;
  if (plan_wgrad(d, &pl)) return 0;
  return (size_t)pl.splitk * pl.K * d->N * sizeof(float);
}

static int conv_wgrad_impl(const s2i_wgrad_desc* d, int planes, const float* a, const float* cvec, const float* g,
                           float* grad_oihw, void* ws, size_t ws_bytes, void* stream, int a16 = 0, int g16 = 0,
                           const float* a_coef = nullptr);

// apply-on-load weight gradient: the generic 128 x 128 fp32 kernel only (the discriminator towers' layers)
static bool wgrad_in_ok(const s2i_wgrad_desc* d, const WgPlan& pl) {
  const int g = d->a_groups < 1 ? 1 : d->a_groups;
  return d->a_act == S2I_ACT_LRELU && d->Cc == 0 && !d->swap && !pl.rows3 && !pl.small_n && pl.tile == 0 && g <= 3 &&
         (d->B % g) == 0;
}

extern "C" int s2i_conv_wgrad_in_eligible(const s2i_wgrad_desc* d) {
  WgPlan pl;
  if (plan_wgrad(d, &pl)) return 0;
  return wgrad_in_ok(d, pl) ? 1 : 0;
}

extern "C" int s2i_conv_wgrad_in(const s2i_wgrad_desc* d, const float* a_raw, const float* a_coef, const float* g,
                                 float* grad_oihw, void* ws, size_t ws_bytes, void* stream) {
  S2I_REQUIRE(a_coef != nullptr, "wgrad(apply-on-load): null coefficient table");
  return conv_wgrad_impl(d, 0, a_raw, nullptr, g, grad_oihw, ws, ws_bytes, stream, 0, 0, a_coef);
}

extern "C" size_t s2i_wgrad_workspace_bytes_dt(const s2i_wgrad_desc* d, int a_dtype, int g_dtype) {
  WgPlan pl;
  if (plan_wgrad(d, &pl, (a_dtype && g_dtype) ? 16 : ((a_dtype || g_dtype) ? 1 : 0))) return 0;
  return (size_t)pl.splitk * pl.K * d->N * sizeof(float);
}

extern "C" int s2i_conv_wgrad_dt(const s2i_wgrad_desc* d, const void* a, int a_dtype, const float* cvec, const void* g,
                                 int g_dtype, float* grad_oihw, void* ws, size_t ws_bytes, void* stream) {
  S2I_REQUIRE((a_dtype == S2I_DT_F32 || a_dtype == S2I_DT_BF16) && (g_dtype == S2I_DT_F32 || g_dtype == S2I_DT_BF16),
              "wgrad: unknown dtype");
  return conv_wgrad_impl(d, 0, (const float*)a, cvec, (const float*)g, grad_oihw, ws, ws_bytes, stream,
                         a_dtype == S2I_DT_BF16, g_dtype == S2I_DT_BF16);
}

extern "C" int s2i_conv_wgrad(const s2i_wgrad_desc* d, const float* a, const float* cvec, const float* g,
                              float* grad_oihw, void* ws, size_t ws_bytes, void* stream) {
  return conv_wgrad_impl(d, 0, a, cvec, g, grad_oihw, ws, ws_bytes, stream);
}

extern "C" size_t s2i_wgrad_workspace_bytes_split(const s2i_wgrad_desc* d, int planes) {
  WgPlan pl;
  if (plan_wgrad(d, &pl, planes)) return 0;
  return (size_t)pl.splitk * pl.K * d->N * sizeof(float);
}

extern "C" int s2i_conv_wgrad_split(const s2i_wgrad_desc* d, int planes, const float* a, const float* cvec,
                                    const float* g, float* grad_oihw, void* ws, size_t ws_bytes, void* stream) {
  S2I_REQUIRE(planes >= 1 && planes <= 3, "wgrad(split): need 1 to 3 bf16 planes");
  return conv_wgrad_impl(d, planes, a, cvec, g, grad_oihw, ws, ws_bytes, stream);
}

template <int BM, int BN, int WM, int WN>
static void launch_wgrad_split(const WgradP& p, dim3 grid, int planes, hipStream_t st) {
  if (planes == 1) hipLaunchKernelGGL((igemm_wgrad_split_kernel<BM, BN, WM, WN, 1>), grid, dim3(256), 0, st, p);
  else if (planes == 2) hipLaunchKernelGGL((igemm_wgrad_split_kernel<BM, BN, WM, WN, 2>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((igemm_wgrad_split_kernel<BM, BN, WM, WN, 3>), grid, dim3(256), 0, st, p);
}

static int conv_wgrad_impl(const s2i_wgrad_desc* d, int planes, const float* a, const float* cvec, const float* g,
                           float* grad_oihw, void* ws, size_t ws_bytes, void* stream, int a16, int g16, const float* a_coef) {
  WgPlan pl;
  // bf16 operands: the plan of the split modes (no row-segment / 96-row tiles, which stage fp32 rows)
  if (plan_wgrad(d, &pl, (a16 && g16) ? 16 : ((planes || a16 || g16) ? 1 : 0))) return 1;
  S2I_REQUIRE(!(planes && (a16 || g16)), "wgrad(split): bf16 tensors go through s2i_conv_wgrad_dt");
  S2I_REQUIRE(!(pl.small_n && g16), "wgrad: the <= 4 channel gradient stream expects an fp32 output gradient");
  S2I_REQUIRE((a || d->Ca == 0) && g && grad_oihw, "wgrad: null operand");
  S2I_REQUIRE(d->Cc == 0 || cvec != nullptr, "wgrad: cvec is null but Cc > 0");
  const size_t need = (size_t)pl.splitk * pl.K * d->N * sizeof(float);
  S2I_REQUIRE(ws && ws_bytes >= need, "wgrad: workspace too small (%zu < %zu)", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  WgradP p;
  p.a = a; p.cvec = cvec; p.g = g; p.slab = (float*)ws;
  p.B = d->B; p.H = d->H; p.W = d->W; p.Ca = d->Ca; p.Cc = d->Cc; p.Cin = pl.Cin;
  p.Ho = pl.Ho; p.Wo = pl.Wo; p.lgWo = s2i_ilog2(pl.Wo); p.lgHoWo = s2i_ilog2(pl.Ho * pl.Wo);
  p.M = pl.M; p.N = d->N; p.ldg = d->ldg; p.K = pl.K; p.T = pl.T; p.kind = d->kind;
  p.cps = pl.cps; p.nchunks = pl.nchunks;
  p.a16 = a16; p.g16 = g16;
  p.a_coef = a_coef; p.a_groups = d->a_groups < 1 ? 1 : d->a_groups; p.a_ipg = d->B / p.a_groups;
  if (a_coef) {
    S2I_REQUIRE(!planes && !a16 && !g16 && wgrad_in_ok(d, pl),
                "wgrad(apply-on-load): fp32 operands, LeakyReLU producer, no broadcast vector, 128 x 128 tile plan (check "
                "s2i_conv_wgrad_in_eligible)");
  }
  {
    const unsigned long long ab = (unsigned long long)d->B * d->H * d->W * d->Ca * (a16 ? 2ull : 4ull);
    const unsigned long long gb = (unsigned long long)pl.M * d->ldg * (g16 ? 2ull : 4ull);
    S2I_REQUIRE(ab < 0x7ff00000ull && gb < 0x7ff00000ull, "wgrad: tensor exceeds the 2 GiB buffer-addressing window");
    p.a_bytes = (unsigned)ab; p.g_bytes = (unsigned)gb;
    p.c_bytes = (unsigned)((unsigned long long)d->B * d->Cc * 4ull);
  }
  dim3 grid(pl.gridK, pl.gridN, pl.splitk);
  if (!a16 && g16 && d->kind == S2I_CONV_K4S2 && d->Ca == 4 && d->Cc == 0 && pl.K == 64 && d->N <= 64 && (d->N % 8) == 0 &&
      (d->ldg % 8) == 0) {
    // first discriminator conv: fp32 NHWC4 image x bf16 output gradient on the bf16 matrix cores
    WgradP q = p;
    q.nchunks = s2i_cdiv(pl.M, 64);
    q.cps = s2i_cdiv(q.nchunks, pl.splitk);
    dim3 g32(1, 1, s2i_cdiv(q.nchunks, q.cps));
    pl.splitk = (int)g32.z;
    pl.gridK = 1; pl.gridN = 1;
    hipLaunchKernelGGL((igemm_wgrad_b16_kernel<64, 64, 2, 2, true>), g32, dim3(256), 0, st, q);
  } else if (a16 && g16 && !pl.small_n && d->Cc == 0 && (pl.Cin % 8) == 0 && (d->N % 8) == 0 && (d->ldg % 8) == 0) {
    // both operands bf16: 64-pixel stages on the bf16 matrix cores
    WgradP q = p;
    q.nchunks = s2i_cdiv(pl.M, 64);
    q.cps = s2i_cdiv(q.nchunks, pl.splitk);
    dim3 g16grid(pl.gridK, pl.gridN, s2i_cdiv(q.nchunks, q.cps));
    S2I_REQUIRE((int)g16grid.z <= pl.splitk, "wgrad(bf16): split plan mismatch");
    // slabs of splits that this plan does not launch must not be summed: shrink the slab count instead
    pl.splitk = (int)g16grid.z;
    if (pl.tile == 7) hipLaunchKernelGGL((igemm_wgrad_b16_kernel<256, 256, 4, 4>), g16grid, dim3(1024), 0, st, q);
    else if (pl.tile == 6) hipLaunchKernelGGL((igemm_wgrad_b16_kernel<256, 128, 4, 2>), g16grid, dim3(512), 0, st, q);
    else if (pl.tile == 0) hipLaunchKernelGGL((igemm_wgrad_b16_kernel<128, 128, 2, 2>), g16grid, dim3(256), 0, st, q);
    else if (pl.tile == 1) hipLaunchKernelGGL((igemm_wgrad_b16_kernel<128, 64, 2, 2>), g16grid, dim3(256), 0, st, q);
    else if (pl.tile == 2) hipLaunchKernelGGL((igemm_wgrad_b16_kernel<128, 32, 4, 1>), g16grid, dim3(256), 0, st, q);
    else hipLaunchKernelGGL((igemm_wgrad_b16_kernel<64, 64, 2, 2>), g16grid, dim3(256), 0, st, q);
  } else if (pl.rows3) {
    dim3 g3(d->Ca == 32 ? 1 : 3, pl.gridN, pl.splitk);
    if (d->Ca == 32 && pl.bn3 == 64) hipLaunchKernelGGL((wgrad_k3_rows_kernel<32, 64, 3, 1, 3>), g3, dim3(192), 0, st, p);
    else if (d->Ca == 32) hipLaunchKernelGGL((wgrad_k3_rows_kernel<32, 32, 3, 1, 3>), g3, dim3(192), 0, st, p);
    else if (pl.bn3 == 128) hipLaunchKernelGGL((wgrad_k3_rows_kernel<64, 128, 2, 2, 1>), g3, dim3(256), 0, st, p);
    else if (pl.bn3 == 64) hipLaunchKernelGGL((wgrad_k3_rows_kernel<64, 64, 2, 2, 1>), g3, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((wgrad_k3_rows_kernel<64, 32, 2, 1, 1>), g3, dim3(128), 0, st, p);
  } else if (planes && !pl.small_n) {
    if (pl.tile == 0) launch_wgrad_split<128, 128, 2, 2>(p, grid, planes, st);
    else if (pl.tile == 1) launch_wgrad_split<128, 64, 2, 2>(p, grid, planes, st);
    else if (pl.tile == 2) launch_wgrad_split<128, 32, 4, 1>(p, grid, planes, st);
    else launch_wgrad_split<64, 64, 2, 2>(p, grid, planes, st);
  } else if (pl.small_n) {
    if (d->Ca == 16) hipLaunchKernelGGL(small_n_wgrad_kernel<4>, dim3(pl.splitk), dim3(256), 0, st, p);
    else if (d->Ca == 32) hipLaunchKernelGGL(small_n_wgrad_kernel<8>, dim3(pl.splitk), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(small_n_wgrad_kernel<16>, dim3(pl.splitk), dim3(256), 0, st, p);
  } else if (pl.tile == 9) hipLaunchKernelGGL((igemm_wgrad_kernel<256, 256, 4, 4>), grid, dim3(1024), 0, st, p);
  else if (pl.tile == 8) hipLaunchKernelGGL((igemm_wgrad_kernel<256, 128, 4, 2>), grid, dim3(512), 0, st, p);
  else if (pl.tile == 0 && a_coef) hipLaunchKernelGGL((igemm_wgrad_kernel<128, 128, 2, 2, true>), grid, dim3(256), 0, st, p);
  else if (pl.tile == 0) hipLaunchKernelGGL((igemm_wgrad_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, p);
  else if (pl.tile == 1) hipLaunchKernelGGL((igemm_wgrad_kernel<128, 64, 2, 2>), grid, dim3(256), 0, st, p);
  else if (pl.tile == 2) hipLaunchKernelGGL((igemm_wgrad_kernel<128, 32, 4, 1>), grid, dim3(256), 0, st, p);
  else if (pl.tile == 3) hipLaunchKernelGGL((igemm_wgrad_kernel<64, 64, 2, 2>), grid, dim3(256), 0, st, p);
  else if (pl.tile == 4) hipLaunchKernelGGL((igemm_wgrad_kernel<96, 64, 3, 1>), grid, dim3(192), 0, st, p);
  else hipLaunchKernelGGL((igemm_wgrad_kernel<96, 32, 3, 1>), grid, dim3(192), 0, st, p);
  S2I_LAUNCH_CHECK("igemm_wgrad");
  {
    const int ncols = d->swap ? d->I : d->O, nrows = d->swap ? d->O : d->I;
    int RT = 8;
    while (RT > 1 && (long long)s2i_cdiv(ncols, 32) * s2i_cdiv(nrows, RT) < 128) RT >>= 1;
    dim3 fgrid(s2i_cdiv(ncols, 32), s2i_cdiv(nrows, RT));
    const size_t shb = (size_t)pl.T * RT * 33 * sizeof(float);
    // two passes (measured: 27 + 17 us against 53 us for one pass that walks the slabs tile by tile): first a
    // float4 stream folds the split slabs into slab 0, then the tile kernel transposes slab 0 into OIHW
    int S = pl.splitk;
    const long long kn = (long long)pl.K * d->N;
    if (pl.small_n || (S >= 32 && (kn % 4) == 0 && kn / 4 < (1 << 18))) {
      // many slabs of a small K x N: the float4 stream below would run on a handful of blocks
      hipLaunchKernelGGL(slab_sum_tree_kernel, dim3(s2i_cdiv(kn / 4, 4)), dim3(256), 0, st, (float*)ws, S, (int)(kn / 4));
      S2I_LAUNCH_CHECK("slab_sum_tree");
      S = 1;
    } else if (S > 2 && (kn % 4) == 0) {
      int sb = s2i_cdiv(kn / 4, 256);
      if (sb > 4096) sb = 4096;
      hipLaunchKernelGGL(slab_sum_kernel, dim3(sb), dim3(256), 0, st, (float*)ws, S, kn / 4);
      S2I_LAUNCH_CHECK("slab_sum");
      S = 1;
    }
    hipLaunchKernelGGL(wgrad_finish_kernel, fgrid, dim3(256), shb, st, (const float*)ws, S, pl.K, d->N,
                       pl.Cin, d->O, d->I, d->KH * d->KW, pl.T, d->swap, d->fold, d->accumulate, RT, grad_oihw,
                       d->i_off, d->I_total > 0 ? d->I_total : d->I);
  }
  S2I_LAUNCH_CHECK("wgrad_finish");
  return 0;
}

extern "C" int s2i_pack_conv_weights_batched(const s2i_pack_item* items_dev, int n, int total_blocks, int max_taps,
                                             void* stream) {
  S2I_REQUIRE(items_dev && n > 0 && total_blocks > 0 && max_taps > 0 && max_taps <= 16, "pack(batched): bad args");
  const size_t shb = (size_t)64 * (8 * max_taps + 1) * sizeof(float);
  hipLaunchKernelGGL(pack_weight_batched_kernel, dim3(total_blocks), dim3(256), shb, (hipStream_t)stream, items_dev, n);
  S2I_LAUNCH_CHECK("pack_weight_batched");
  return 0;
}

extern "C" int s2i_pack_conv_weight(const float* w_oihw, float* packed, int O, int I, int KH, int KW, int Ip,
                                    int mode, void* stream) {
  S2I_REQUIRE(w_oihw && packed, "pack: null pointer");
  S2I_REQUIRE(O > 0 && I > 0 && KH > 0 && KW > 0 && Ip >= I, "pack: bad shape");
  int T = KH * KW;
  if (mode == S2I_PACK_UPFOLD) {
    S2I_REQUIRE(KH == 3 && KW == 3, "pack: UPFOLD needs a 3x3 parameter");
    T = 16;
  } else {
    S2I_REQUIRE(mode == S2I_PACK_PLAIN, "pack: unknown mode %d", mode);
  }
  const int Op = (O + 3) & ~3;
  {
    dim3 pgrid(s2i_cdiv(Op, 64), s2i_cdiv(Ip, 8));
    const size_t shb = (size_t)64 * (8 * KH * KW + 1) * sizeof(float);
    hipLaunchKernelGGL(pack_weight_kernel, pgrid, dim3(256), shb, (hipStream_t)stream, w_oihw, packed, O, I, KH * KW,
                       Ip, Op, T, mode);
  }
  S2I_LAUNCH_CHECK("pack_weight");
  return 0;
}
