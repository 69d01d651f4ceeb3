// bf16 activation path (BASELINE config 4): implicit-GEMM convolutions whose activations live in HBM as bf16 NHWC,
// weights as bf16 (fp32 masters stay in the trainer's flat buffers), products on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, BatchNorm statistics taken from the fp32 accumulators.
//
// What is different from the fp32 / split kernels of s2i_igemm.hip (which gather an im2col chunk per 32-deep K step
// from global memory, i.e. pull every input pixel through L2 once per tap):
//   * the block stages a 2-D input PATCH with its halo in LDS once per channel chunk -- the pixels of TB images x
//     (TH-1)*s+KH rows x (TW-1)*s+KW columns -- and every tap reads its MFMA A-fragments from that patch at a
//     wave-uniform offset.  Out-of-image halo pixels are zero-filled by the bounds-checked buffer load, so the
//     matrix loop has no masks at all;
//   * the K loop runs (channel chunk, tap group) instead of (tap, channel): one LDS stage carries all 9 taps of a
//     3x3 (8 of the 16 taps of a 4x4, the 4 taps of one transposed-conv phase) for CK channels, 32-36 MFMAs per wave
//     between two barriers instead of 8;
//   * weights are pre-arranged at pack time in exactly the order the stages consume them,
//     Wb[phase][chunk][tap][n][CK] (tap flip / transposition / parity tap selection of the input-gradient forms
//     already applied), so a stage's B tile is ONE contiguous run of global memory;
//   * LDS rows are CK*2 bytes (32 / 64 / 128) with the 16-byte segments XOR-swizzled by the row index, so a
//     ds_read_b128 fragment read is conflict-free without padding;
//   * bf16 results leave through an LDS transpose and 16-byte row-contiguous stores.
// Rows of the GEMM are output pixels; a block owns a tile of 128 of them shaped TB x TH x TW (powers of two) chosen
// from the map size, e.g. 4 x 32 pixels of one image on wide maps, 8 whole 4x4 maps at the discriminators' tails.
#include "s2i_common.h"
#include <stdio.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define S2I_OOB 0x7ffffff0

enum { KB_K3S1 = 0, KB_K4S2 = 1, KB_TCONV = 2 };

struct ConvBP {
  const unsigned short* __restrict__ x;
  const unsigned short* __restrict__ w;
  const float* __restrict__ cls_bias;
  unsigned short* __restrict__ y;
  float* __restrict__ part;
  float* __restrict__ slab;
  int B, H, W, C;
  int Ho, Wo;              // output grid of one phase
  int N, Npad, ldy;
  int lgTW, lgTH, lgTB;
  int tilesX, tilesY;      // tiles per image along x / y (powers of two)
  int ntiles;              // tiles in all (tilesX * tilesY * image groups)
  int PH, PW, npix;
  int nchunk, splitk, cps;
  int stats, nparts;
  long long Mrows;         // rows of y (all phases)
  unsigned x_bytes, w_bytes;
  int dbg;                 // ablation switches of tools/conv16_bench.py (S2I_B16_DBG; 0 in production): 1 no patch loads after
                           // the first, 2 no weight loads after the first, 4 no LDS staging after the first, 8 no MFMA,
                           // 16 no output stores
};

__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// KIND: geometry; BN: output channels per block; CK: channels per LDS stage; waves 2x2 (BN >= 64) or 4x1 (BN = 32)
// TG: taps per LDS stage (T / TG stages per channel chunk); fragment reads are software-pipelined one k-step ahead with the
// order pinned (sched_barrier).  (Measured and removed: compiler-scheduled fragment reads, fewer taps per stage at three
// blocks per CU, weights by LDS-DMA into two buffers with one barrier per stage -- all within 5 %)
template <int KIND, int BN, int CK, int TG, int DBG = 0>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(ConvBP p) {
  constexpr int T = KIND == KB_K3S1 ? 9 : (KIND == KB_K4S2 ? 16 : 4);
  constexpr int NG = T / TG;
  static_assert(NG * TG == T, "tap groups must tile the taps");
  constexpr int WAVES_N = BN >= 64 ? 2 : 1, WAVES_M = 4 / WAVES_N;
  constexpr int TM = 128 / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr int SEGS = CK / 8;                      // 16-byte segments per LDS row
  constexpr int ROWB = CK * 2;
  constexpr int LGR = SEGS == 2 ? 3 : (SEGS == 4 ? 2 : 1);  // rows per 256-byte bank span = 16 / SEGS -> swizzle shift
  constexpr int KS = CK / 16;                       // k-steps per tap
  constexpr int BSEG = TG * BN * SEGS;              // 16-byte segments of one weight stage
  constexpr int NBL = (BSEG + 255) / 256;
  constexpr int MAXPIX = KIND == KB_K4S2 ? (CK == 16 ? 800 : 672) : 320;  // plan_bf16 checks the patch against this
  constexpr int NPL = (MAXPIX * SEGS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [patch | weight stage], reused by the epilogue
  unsigned char* As = smem;
  unsigned char* Bs = smem + ((p.npix * ROWB + 255) & ~255);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware block order (round 3).  The gridDim.y channel blocks and the gridDim.z phases / K splits of ONE pixel tile
  // read the same input patch; blocks are dealt round-robin over the 8 XCDs (an L2 each), and in launch order the
  // siblings of a tile are gridDim.x ids apart: another XCD, another time.  Linear id L -> (tile, sibling) such that all
  // siblings of a tile have the same L % 8 and consecutive ids on that XCD (a bijection: the last group of tiles uses
  // its own modulus).
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
  if (KIND == KB_TCONV) { phase = bz / p.splitk; split = bz - phase * p.splitk; }
  const int py = phase >> 1, px = phase & 1;
  const int n0 = by * BN;
  const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
  const int tix = bx & (p.tilesX - 1);
  const int tiy = (bx / p.tilesX) & (p.tilesY - 1);
  const int tib = bx / (p.tilesX * p.tilesY);
  const int b0 = tib << p.lgTB, oy0 = tiy << p.lgTH, ox0 = tix << p.lgTW;
  const int PW = p.PW, PH = p.PH;
  // input coordinate of patch pixel (0, 0)
  int iy0, ix0;
  if (KIND == KB_K3S1) { iy0 = oy0 - 1; ix0 = ox0 - 1; }
  else if (KIND == KB_K4S2) { iy0 = 2 * oy0 - 1; ix0 = 2 * ox0 - 1; }
  else { iy0 = oy0 - (py ? 0 : 1); ix0 = ox0 - (px ? 0 : 1); }

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // DBG & 32 (diagnostic build of tools/conv16_timeline.py only): wave 0 stamps s_memtime at the phase boundaries into a
  // buffer of its own (p.slab, 64 stamps per block); no stamp executes in the production instantiation
  auto stamp = [&](int i) {
    if constexpr ((DBG & 32) != 0) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (tid == 0) {
        const size_t blk = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        reinterpret_cast<unsigned long long*>(p.slab)[blk * 64 + i] = t;
      }
    }
  };
  stamp(0);
  // ---- patch staging plan of this thread: global byte offset (chunk 0) and swizzled LDS offset per load ----
  int pgo[NPL], plo[NPL];
  const int nseg = p.npix * SEGS;
  const float inv_pw = 1.0f / (float)PW, inv_ph = 1.0f / (float)PH;
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    const int e = tid + q * 256;
    int go = S2I_OOB, lo = -1;
    if (e < nseg) {
      // pix < 1024: floor(pix / PW) = (int)((pix + 0.5) * (1 / PW)) exactly in fp32 (no integer-division sequences)
      const int pix = e / SEGS, seg = e & (SEGS - 1);
      const int rest = (int)(((float)pix + 0.5f) * inv_pw);
      const int xl = pix - rest * PW;
      const int tb = (int)(((float)rest + 0.5f) * inv_ph);
      const int yl = rest - tb * PH;
      const int b = b0 + tb, iy = iy0 + yl, ix = ix0 + xl;
      if (b < p.B && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) go = (((b * p.H + iy) * p.W + ix) * p.C + seg * 8) * 2;
      const int xs = KIND == KB_K4S2 ? (xl & 1) * (PW >> 1) + (xl >> 1) : xl;   // even / odd columns de-interleaved
      const int prow = (tb * PH + yl) * PW + xs;
      lo = prow * ROWB + ((seg ^ ((prow >> LGR) & (SEGS - 1))) << 4);
    }
    pgo[q] = go;
    plo[q] = lo;
  }
  // ---- A fragment rows of this lane: patch row of (tile row, tap (0,0)) ----
  int arow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * TM * 32 + i * 32 + l31;
    const int tx = r & (TW - 1), ty = (r >> p.lgTW) & (TH - 1), tb = r >> (p.lgTW + p.lgTH);
    arow[i] = (tb * PH + (KIND == KB_K4S2 ? 2 * ty : ty)) * PW + tx;
  }
  // weight fragment rows of this lane inside one tap's [BN][CK] image (BN is a multiple of the swizzle period, so the
  // tap only adds tl * BN * ROWB)
  int boffs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int brow = wn * TN * 32 + j * 32 + l31;
    boffs[j] = brow * ROWB + ((lh ^ ((brow >> LGR) & (SEGS - 1))) << 4);
  }
  // weight stage: segment e -> (tap, n, seg); global element offset inside the stage and swizzled LDS offset
  // (computed on the fly: BN and SEGS are powers of two)
  const int wstage = TG * p.Npad * CK;               // elements per (chunk, tap group)
  u32x4 ra[NPL], rb[NBL];
  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunk, c_begin + p.cps);
  const int s_begin = c_begin * NG, s_end = c_end * NG;

  auto fetch = [&](int st, bool with_a) {            // st = chunk * NG + group
    const int cc = st / NG, tg = st - cc * NG;
    if (with_a && !((DBG & 1) && st != s_begin)) {
      const int coff = cc * CK * 2;
#pragma unroll
      for (int q = 0; q < NPL; ++q) ra[q] = bload16(rx, pgo[q] == S2I_OOB ? S2I_OOB : pgo[q] + coff);
    }
    const int wbase = ((phase * p.nchunk + cc) * NG + tg) * wstage + n0 * CK;
    if ((DBG & 2) && st != s_begin) return;
    {
#pragma unroll
      for (int q = 0; q < NBL; ++q) {
        const int e = tid + q * 256;
        const int seg = e & (SEGS - 1), n = (e / SEGS) & (BN - 1), t = e / (SEGS * BN);
        rb[q] = bload16(rw, e < BSEG ? (wbase + (t * p.Npad + n) * CK + seg * 8) * 2 : S2I_OOB);
      }
    }
  };
  auto stage = [&](bool with_a) {
    if (with_a) {
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        if (plo[q] >= 0) *reinterpret_cast<u32x4*>(As + plo[q]) = ra[q];
    }
    {
#pragma unroll
      for (int q = 0; q < NBL; ++q) {
        const int e = tid + q * 256;
        if (e < BSEG) {
          const int seg = e & (SEGS - 1), row = e / SEGS;  // row = t * BN + n
          *reinterpret_cast<u32x4*>(Bs + row * ROWB + ((seg ^ ((row >> LGR) & (SEGS - 1))) << 4)) = rb[q];
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

  if (s_begin < s_end) fetch(s_begin, true);
  stamp(1);
  for (int st = s_begin; st < s_end; ++st) {
    const bool new_a = (st % NG) == 0;
    const int sb = 2 + 6 * min(st - s_begin, 7);
    stamp(sb);
    if (!((DBG & 4) && st != s_begin)) stage(new_a);
    stamp(sb + 1);
    __syncthreads();
    stamp(sb + 2);
    if (st + 1 < s_end) fetch(st + 1, ((st + 1) % NG) == 0);
    const int tg = st % NG;
    const unsigned char* Bcur = Bs;
    // LDS byte offsets of this lane's fragments, once per stage: the matrix loop below then spends no vector
    // instructions on addresses (row, swizzle and tap arithmetic per read had the address math competing with the MFMAs
    // for the SIMD's issue slots).  k-step ks of a tap is the ks = 0 offset XOR (ks << 5): the segment index is
    // (2 ks) ^ lh ^ swizzle(row).
    int aaddr[TM][TG];
#pragma unroll
    for (int tl = 0; tl < TG; ++tl) {
      const int t = tg * TG + tl;
      int toff;  // patch rows between tap (0,0) and tap t
      if (KIND == KB_K3S1) { toff = (t / 3) * PW + (t % 3); }
      else if (KIND == KB_K4S2) { const int dy = t >> 2, dx = t & 3; toff = dy * PW + (dx & 1) * (PW >> 1) + (dx >> 1); }
      else { const int ta = t >> 1, tb2 = t & 1; toff = (py ? ta : 1 - ta) * PW + (px ? tb2 : 1 - tb2); }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int prow = arow[i] + toff;
        aaddr[i][tl] = prow * ROWB + ((lh ^ ((prow >> LGR) & (SEGS - 1))) << 4);
      }
    }
    // fragments of k-step s+1 are read from LDS before the MFMAs of k-step s are issued (two named register sets,
    // order pinned): the LDS latency hides behind this wave's own MFMAs, and no more than two sets are ever live
    auto ldfr = [&](int stp, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
      const int tl = stp / KS, ks = stp % KS;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + (aaddr[i][tl] ^ (ks << 5)));
#pragma unroll
      for (int j = 0; j < TN; ++j)
        b[j] = *reinterpret_cast<const bf16x8*>(Bcur + tl * BN * ROWB + (boffs[j] ^ (ks << 5)));
    };
    auto mma = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if constexpr (!(DBG & 8)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          else asm volatile("" :: "v"(a[i]), "v"(b[j]));  // keeps the fragment reads alive
    };
    constexpr int NS = TG * KS;
    stamp(sb + 3);
    {
      bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
      ldfr(0, a0, b0);
#pragma unroll
      for (int s2 = 0; s2 < NS; s2 += 2) {
        if (s2 + 1 < NS) ldfr(s2 + 1, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s2 + 2 < NS) ldfr(s2 + 2, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s2 + 1 < NS) mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stamp(sb + 4);
    __syncthreads();
    stamp(sb + 5);
  }

  // ---- epilogue ----
  const bool raw = p.splitk > 1;
  // global row (pixel of y) of tile row r, or -1 beyond the batch
  auto out_row = [&](int r) -> long long {
    const int tx = r & (TW - 1), ty = (r >> p.lgTW) & (TH - 1), tb = r >> (p.lgTW + p.lgTH);
    const int b = b0 + tb, oy = oy0 + ty, ox = ox0 + tx;
    if (b >= p.B) return -1;
    if (KIND == KB_TCONV) return ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
    return ((long long)b * p.Ho + oy) * p.Wo + ox;
  };
  if (p.cls_bias && !raw) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int tx = rr & (TW - 1), ty = (rr >> p.lgTW) & (TH - 1), tb = rr >> (p.lgTW + p.lgTH);
        const int b = b0 + tb, oy = oy0 + ty, ox = ox0 + tx;
        if (b >= p.B) continue;
        const int cls = 3 * (oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1)) + (ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1));
        const float* bp = p.cls_bias + ((size_t)b * 9 + cls) * p.N;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * TN * 32 + j * 32 + l31;
          if (n < p.N) acc[i][j][r] += bp[n];
        }
      }
  }
  if (raw) {
    float* outp = p.slab + (size_t)split * p.Mrows * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = out_row(wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
        if (row < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * TN * 32 + j * 32 + l31;
          if (n < p.N) outp[row * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  // bf16 tile through LDS: lanes l and l^1 (columns c, c+1) exchange one register of each row pair so that every lane
  // owns two adjacent columns of ONE row and writes them as a dword; rows leave as 16-byte row-contiguous stores
  constexpr int ERS = BN * 2 + 16;
  const bool odd = lane & 1;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float mine0 = acc[i][j][2 * q], mine1 = acc[i][j][2 * q + 1];
        const float give = odd ? mine0 : mine1;
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
        const int reg = 2 * q + (odd ? 1 : 0);
        const int rr = wm * TM * 32 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
        const int col = wn * TN * 32 + j * 32 + (l31 & ~1);
        const unsigned v = odd ? pack2(got, mine1) : pack2(mine0, got);
        *reinterpret_cast<unsigned*>(smem + rr * ERS + col * 2) = v;
      }
  __syncthreads();
  stamp(50);
  {
    constexpr int SPR = BN / 8;                       // 16-byte segments per row
#pragma unroll
    for (int q = 0; q < 128 * SPR / 256; ++q) {
      const int e = tid + q * 256;
      const int rr = e / SPR, sg = e & (SPR - 1);
      const long long row = out_row(rr);
      const int n = n0 + sg * 8;
      if (row >= 0 && n < p.N && !(DBG & 16))
        *reinterpret_cast<u32x4*>(p.y + row * p.ldy + n) = *reinterpret_cast<const u32x4*>(smem + rr * ERS + sg * 16);
    }
  }
  stamp(51);
  if (p.stats) {
    // column sums of the fp32 accumulators over this block's valid rows (rows beyond the batch gathered zeros)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);  // [2][WAVES_M][BN]
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
  stamp(52);
  if constexpr ((DBG & 32) != 0) {
    if (tid == 0) {
      const size_t blk = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.slab) + blk * 64;
      o[60] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
      o[61] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
      o[62] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// ---- second-generation kernel: 256 output pixels x 128 channels per tile, 8 waves, persistent over tiles ---------------
// What the in-kernel timelines showed (tools/conv16_timeline.py).  128-pixel kernel: the matrix loop
// was a quarter of a block's lifetime; the rest was the bulk issue of the next stage's loads (the CU's vector-memory path
// stalls the issuing wave), waiting for loads issued one matrix phase earlier, two barriers per stage, and a prologue /
// epilogue per 33 MFLOP.  First 256-pixel version (everything prefetched, one barrier per stage, one block per CU): the
// steady-state stage ran the matrix pipe at 82 %, but 37 % of a block's lifetime was its prologue -- every CU of the chip
// starts a tile at the same moment and asks HBM for its first patch chunk at once (36 MB per round: 7 us at 5 TB/s), then
// leaves HBM idle while it computes.  This kernel therefore
//   * keeps the 256-pixel tile (the 256 KB weight stream of a tile is amortised over twice the pixels),
//   * issues every global load BETWEEN matrix instructions, a few per k-step: weights two stages ahead (two register
//     sets that alternate), the next patch chunk a whole chunk ahead; the LDS stores of a later stage are interleaved too,
//   * keeps two weight stages (and two patch chunks where the patch is small: 3x3, transposed phases) in LDS: ONE barrier
//     per stage,
//   * is PERSISTENT where the launcher gives it fewer blocks than tiles (stride-2 4x4): the first patch chunk and the
//     first weight stages of the NEXT tile are prefetched during the last chunk of the current one exactly like any
//     other next chunk, so HBM is asked for data all the time instead of in bursts, and a tile boundary costs the
//     epilogue plus one patch store.
// The schedule is compile-time: the stages of a channel chunk are unrolled (tap group tg, weight register set parity P),
// so every vmcnt the compiler derives is exact; loads that must move nothing go through a descriptor of zero records.
// Geometry, LDS images, swizzle, fragment maps and the epilogue are those of conv_bf16_kernel.  A patch segment is
// described by its byte offset from the patch's pixel (0,0) plus four halo flags (left / right column, top / bottom
// row) that are tested against the tile's position, so moving to the next tile costs no per-lane index arithmetic.
template <int V> struct IC { static constexpr int value = V; };

// PWC != 0: the patch is PWC pixels wide (compile time) and its LDS rows are padded to 80 bytes instead of XOR-swizzled: a
// fragment address is then lane base + constant, i.e. the matrix loop spends NO vector instruction on addresses (with
// one k-step of read-ahead, a dependent address chain in front of every read starved the matrix pipe: +40 % per stage).
// 80 = 5 x 16 bytes and 5 is odd, so the 16 lanes of a ds_read_b128 group (consecutive patch rows) hit 16 distinct
// 16-byte bank groups.
template <int KIND, int CK, int TG, bool PDB, int DBG = 0, int PWC = 0>
__global__ __launch_bounds__(512, 1) void conv_bf16_v2_kernel(ConvBP p) {
  constexpr int BN = 128, BM = 256, NT = 512;
  constexpr int T = KIND == KB_K3S1 ? 9 : (KIND == KB_K4S2 ? 16 : 4);
  constexpr int NG = T / TG;
  static_assert(NG * TG == T && NG >= 2, "tap groups must tile the taps, at least two stages per channel chunk");
  constexpr int WAVES_N = 2, WAVES_M = 4, TM = 2, TN = 2;
  constexpr int SEGS = CK / 8, ROWB = CK * 2;
  constexpr bool PADA = PWC != 0;
  constexpr int AROWB = PADA ? ROWB + 16 : ROWB;              // bytes per patch row in LDS
  static_assert(!PADA || (KIND == KB_K4S2 && CK == 32), "padded patch rows: stride-2 4x4 with 32-channel chunks");
  constexpr int LGR = SEGS == 2 ? 3 : (SEGS == 4 ? 2 : 1);
  constexpr int KS = CK / 16, NS = TG * KS, HS = NS / 2;      // k-steps per stage; the first HS issue loads, the rest store
  static_assert((NS % 2) == 0, "even number of k-steps per stage");
  constexpr int BSEG = TG * BN * SEGS, NBL = BSEG / NT, B_BYTES = TG * BN * ROWB;
  static_assert(BSEG % NT == 0, "a weight stage is a whole number of 16-byte segments per thread");
  constexpr int MAXPIX = KIND == KB_K4S2 ? 1280 : (KIND == KB_K3S1 ? 416 : 336);   // plan_bf16 checks the patch against this
  constexpr int NPL = (MAXPIX * SEGS + NT - 1) / NT;
  static_assert(NPL <= 16, "halo flags of a thread's patch segments fit two registers");
  constexpr int ALS = NG - 1;                                 // stages of a chunk that issue the next chunk's patch loads
  constexpr int AG = ALS * HS;                                // ... over this many k-step slots
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [patch (x2) | weight stage x2 | 1 KB sink]
  const int A_BYTES = (p.npix * AROWB + 255) & ~255;
  unsigned char* As = smem;
  unsigned char* Bs = smem + (PDB ? 2 : 1) * A_BYTES;
  const int sink = (PDB ? 2 : 1) * A_BYTES + 2 * B_BYTES;     // lanes without a patch segment store here (no exec masks)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5;
  int phase = 0, split = blockIdx.z;
  if (KIND == KB_TCONV) { phase = blockIdx.z / p.splitk; split = blockIdx.z - phase * p.splitk; }
  const int py = phase >> 1, px = phase & 1;
  const int n0 = blockIdx.y * BN;
  const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
  const int PW = PADA ? PWC : p.PW, PH = p.PH;
  const int ntiles = p.ntiles;

  // DBG & 32 (diagnostic build of tools/conv16_timeline.py): wave 0 stamps s_memtime into a buffer of its own
  auto stamp = [&](int i) {
    if constexpr ((DBG & 32) != 0) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (tid == 0 && i < 64) {
        const size_t blk = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        reinterpret_cast<unsigned long long*>(p.slab)[blk * 64 + i] = t;
      }
    }
  };
  stamp(0);

  // position of a tile: first image / output row / output column, byte offset of the patch's pixel (0,0) in x (may be
  // negative: the halo of the first image), and which of the patch's halo sides lie outside the image
  struct TilePos { int b0, oy0, ox0, base; unsigned edge; };
  auto tile_pos = [&](int tile) -> TilePos {
    TilePos t;
    const int tix = tile & (p.tilesX - 1);
    const int tiy = (tile / p.tilesX) & (p.tilesY - 1);
    const int tib = tile / (p.tilesX * p.tilesY);
    t.b0 = tib << p.lgTB; t.oy0 = tiy << p.lgTH; t.ox0 = tix << p.lgTW;
    int iy0, ix0;
    if (KIND == KB_K3S1) { iy0 = t.oy0 - 1; ix0 = t.ox0 - 1; }
    else if (KIND == KB_K4S2) { iy0 = 2 * t.oy0 - 1; ix0 = 2 * t.ox0 - 1; }
    else { iy0 = t.oy0 - (py ? 0 : 1); ix0 = t.ox0 - (px ? 0 : 1); }
    t.base = (((t.b0 * p.H + iy0) * p.W + ix0) * p.C) * 2;
    t.edge = (t.ox0 == 0 ? 1u : 0u) | (t.ox0 + TW == p.Wo ? 2u : 0u) | (t.oy0 == 0 ? 4u : 0u) | (t.oy0 + TH == p.Ho ? 8u : 0u);
    return t;
  };

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // patch plan of this thread: byte offset of segment q from the patch's pixel (0,0), its halo flags (4 bits per q) and
  // its swizzled LDS offset.  The LDS offsets are kept in registers only where the matrix loop needs them (two patch
  // buffers); the single-buffer form recomputes them at the chunk boundary.
  const int nseg = p.npix * SEGS;
  const float inv_pw = 1.0f / (float)PW, inv_ph = 1.0f / (float)PH;
  int zv = 0;   // zero the optimiser cannot see through (re-made per chunk): keeps per-lane index arithmetic that is invariant
                // across chunks and tiles INSIDE the loops instead of hoisted into registers the matrix loop needs
  auto patch_slot = [&](int q, int& rel, unsigned& flags, int& lo) {
    const int e = tid + zv + q * NT;
    rel = S2I_OOB;
    flags = 0;
    lo = -1;
    if (e < nseg) {
      const int pix = e / SEGS, seg = e & (SEGS - 1);
      const int rest = (int)(((float)pix + 0.5f) * inv_pw);
      const int xl = pix - rest * PW;
      const int tb = (int)(((float)rest + 0.5f) * inv_ph);
      const int yl = rest - tb * PH;
      rel = (((tb * p.H + yl) * p.W + xl) * p.C + seg * 8) * 2;
      // halo sides: 3x3 and stride-2 4x4 (pad 1) have all four; a transposed-conv phase has the top row / left column
      // when its parity is 0 and the bottom row / right column when it is 1
      const bool hl = KIND == KB_TCONV ? px == 0 : true, hr = KIND == KB_TCONV ? px == 1 : true;
      const bool ht = KIND == KB_TCONV ? py == 0 : true, hb = KIND == KB_TCONV ? py == 1 : true;
      flags = (hl && xl == 0 ? 1u : 0u) | (hr && xl == PW - 1 ? 2u : 0u) | (ht && yl == 0 ? 4u : 0u) | (hb && yl == PH - 1 ? 8u : 0u);
      const int xs = KIND == KB_K4S2 ? (xl & 1) * (PW >> 1) + (xl >> 1) : xl;
      const int prow = (tb * PH + yl) * PW + xs;
      lo = PADA ? prow * AROWB + seg * 16 : prow * ROWB + ((seg ^ ((prow >> LGR) & (SEGS - 1))) << 4);
    }
  };
  // (the single-buffer form -- the stride-2 4x4, whose patch is the largest -- recomputes the LDS offsets at the chunk
  // boundary instead of keeping them: its matrix loop has no register to spare, and a spilled value comes back through a
  // scratch load whose vmcnt wait drains the whole prefetch queue)
  int rel[NPL], plo[PDB ? NPL : 1];
  unsigned fw0 = 0, fw1 = 0;
#pragma unroll
  for (int q = 0; q < NPL; ++q) {
    int r, lo;
    unsigned f;
    patch_slot(q, r, f, lo);
    rel[q] = r;
    if (q < 8) fw0 |= f << (4 * q);
    else fw1 |= f << (4 * (q - 8));
    if (PDB) plo[q] = lo;
  }
  auto plo_of = [&](int q) -> int {
    if (PDB) return plo[q];
    int r, lo;
    unsigned f;
    patch_slot(q, r, f, lo);
    return lo;
  };
  // vector offset of patch segment q for a tile at (base, edge): out of range where the segment is halo outside the image
  auto avoff = [&](int q, int base, unsigned edge) -> int {
    const unsigned hit = (q < 8 ? fw0 : fw1) & (edge << (4 * (q & 7)));
    return hit ? S2I_OOB : base + rel[q];
  };
  auto adst = [&](unsigned char* base, int lo) -> u32x4* {
    return reinterpret_cast<u32x4*>(lo >= 0 ? base + lo : smem + sink + lane * 16);
  };
  // A fragment rows of this lane (tile row -> patch row of tap (0,0)); the same for every tile
  int arow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * TM * 32 + i * 32 + l31;
    const int tx = r & (TW - 1), ty = (r >> p.lgTW) & (TH - 1), tb = r >> (p.lgTW + p.lgTH);
    arow[i] = (tb * PH + (KIND == KB_K4S2 ? 2 * ty : ty)) * PW + tx;
    if (PADA) arow[i] = arow[i] * AROWB + lh * 16;           // padded rows: the lane's byte base, taps and k-steps add constants
  }
  int boffs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int brow = wn * TN * 32 + j * 32 + l31;
    boffs[j] = brow * ROWB + ((lh ^ ((brow >> LGR) & (SEGS - 1))) << 4);
  }
  // weight stage segments of this thread: segment q is row (tid / SEGS) + q * (NT / SEGS) of the stage's [TG * BN][CK]
  // image, so both its global offset and its swizzled LDS offset are the q = 0 values plus a constant
  static_assert(BN % (NT / SEGS) == 0 && ((NT / SEGS) >> LGR) % SEGS == 0, "weight rows per load step keep the swizzle");
  const int bseg = tid & (SEGS - 1), brow0 = tid / SEGS;
  const int bgo0 = (brow0 * CK + bseg * 8) * 2;
  const int blo0 = brow0 * ROWB + ((bseg ^ ((brow0 >> LGR) & (SEGS - 1))) << 4);
  auto bgo_s = [&](int q) -> int {   // uniform part of the global byte offset of segment q
    const int row = q * (NT / SEGS), n = row & (BN - 1), t = row / BN;
    return ((t * p.Npad + n) * CK) * 2;
  };
  constexpr int BLO_STEP = (NT / SEGS) * ROWB;
  const int wstage = TG * p.Npad * CK;                        // elements per (chunk, tap group)
  const int c_begin = split * p.cps;
  const int c_end = min(p.nchunk, c_begin + p.cps);
  const int ncl = c_end - c_begin;                            // chunks of this block's K range
  auto wbyte = [&](int cc, int tg) -> int { return (((phase * p.nchunk + cc) * NG + tg) * wstage + n0 * CK) * 2; };
  // loads: the per-lane offset goes in the vector offset (out of range where nothing is to be loaded), the uniform part in
  // the scalar offset
  auto bload16s = [&](__amdgpu_buffer_rsrc_t r, int voff, int soff) -> u32x4 {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  };

  u32x4 ra[NPL], rb[2][NBL];
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();
  stamp(1);

  int tile = blockIdx.x;
  if (tile >= ntiles || ncl <= 0) return;
  TilePos cur = tile_pos(tile);
  // ---- prologue: patch chunk c_begin of the first tile, weight stages 0 (-> LDS) and 1 (stays in its register set) ----
  {
    const int coff = c_begin * CK * 2;
#pragma unroll
    for (int q = 0; q < NPL; ++q) ra[q] = bload16s(rx, avoff(q, cur.base, cur.edge), coff);
    const int w0 = wbyte(c_begin, 0), w1 = wbyte(c_begin, 1);
#pragma unroll
    for (int q = 0; q < NBL; ++q) rb[0][q] = bload16s(rw, bgo0, w0 + bgo_s(q));
#pragma unroll
    for (int q = 0; q < NBL; ++q) rb[1][q] = bload16s(rw, bgo0, w1 + bgo_s(q));
    stamp(2);
#pragma unroll
    for (int q = 0; q < NPL; ++q) *adst(As, plo_of(q)) = ra[q];
#pragma unroll
    for (int q = 0; q < NBL; ++q) *reinterpret_cast<u32x4*>(Bs + blo0 + q * BLO_STEP) = rb[0][q];
  }
  __syncthreads();
  stamp(3);

  // ---- epilogue of one tile: accumulators -> y (bf16, through an LDS transpose) or fp32 slabs, BatchNorm column sums ----
  auto epilogue = [&](const TilePos& tp, int tile_id) {
    // the lane / wave coordinates are re-derived from a value the optimiser cannot see through, so that none of the
    // epilogue's (tile-invariant) address arithmetic is hoisted above the tile loop into registers the matrix loop needs
    int z = 0;
    asm volatile("" : "+v"(z));
    const int tid = (int)threadIdx.x + z, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lh = lane >> 5;
    const bool raw = p.splitk > 1;
    auto out_row = [&](int r) -> long long {
      const int tx = r & (TW - 1), ty = (r >> p.lgTW) & (TH - 1), tb = r >> (p.lgTW + p.lgTH);
      const int b = tp.b0 + tb, oy = tp.oy0 + ty, ox = tp.ox0 + tx;
      if (b >= p.B) return -1;
      if (KIND == KB_TCONV) return ((long long)b * (2 * p.Ho) + 2 * oy + py) * (2 * p.Wo) + 2 * ox + px;
      return ((long long)b * p.Ho + oy) * p.Wo + ox;
    };
    if (p.cls_bias && !raw) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int tx = rr & (TW - 1), ty = (rr >> p.lgTW) & (TH - 1), tb = rr >> (p.lgTW + p.lgTH);
          const int b = tp.b0 + tb, oy = tp.oy0 + ty, ox = tp.ox0 + tx;
          if (b >= p.B) continue;
          const int cls = 3 * (oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1)) + (ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1));
          const float* bp = p.cls_bias + ((size_t)b * 9 + cls) * p.N;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + l31;
            if (n < p.N) acc[i][j][r] += bp[n];
          }
        }
    }
    if (raw) {
      float* outp = p.slab + (size_t)split * p.Mrows * p.N;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long row = out_row(wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
          if (row < 0) continue;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * TN * 32 + j * 32 + l31;
            if (n < p.N) outp[row * p.N + n] = acc[i][j][r];
          }
        }
      return;
    }
    constexpr int ERS = BN * 2 + 16;
    const bool odd = lane & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float mine0 = acc[i][j][2 * q], mine1 = acc[i][j][2 * q + 1];
          const float give = odd ? mine0 : mine1;
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xF, 0xF, true));
          const int reg = 2 * q + (odd ? 1 : 0);
          const int rr = wm * TM * 32 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
          const int col = wn * TN * 32 + j * 32 + (l31 & ~1);
          const unsigned v = odd ? pack2(got, mine1) : pack2(mine0, got);
          *reinterpret_cast<unsigned*>(smem + rr * ERS + col * 2) = v;
        }
    __syncthreads();
    stamp(50);
    {
      constexpr int SPR = BN / 8;
#pragma unroll
      for (int q = 0; q < BM * SPR / NT; ++q) {
        const int e = tid + q * NT;
        const int rr = e / SPR, sg = e & (SPR - 1);
        const long long row = out_row(rr);
        const int n = n0 + sg * 8;
        if (row >= 0 && n < p.N)
          *reinterpret_cast<u32x4*>(p.y + row * p.ldy + n) = *reinterpret_cast<const u32x4*>(smem + rr * ERS + sg * 16);
      }
    }
    stamp(51);
    if (p.stats) {
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);  // [2][WAVES_M][BN]
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
          const int gm = phase * ntiles + tile_id;
          p.part[((size_t)0 * p.nparts + gm) * p.N + n] = sv;
          p.part[((size_t)1 * p.nparts + gm) * p.N + n] = sq;
        }
      }
    }
    stamp(52);
  };

  // one channel chunk = NG stages, unrolled.  CP: parity of the chunk's first stage among the block's stages.
  // ci: index of the chunk inside [c_begin, c_end).  The chunk AFTER the last one of a tile is the first chunk of the
  // block's next tile (patch and weights alike); after the last tile nothing follows and the loads move nothing.
  int apar = 0;                                               // patch buffer in use (two-buffer form)
  auto run_chunk = [&](auto cp_tag, int ci, const TilePos& nxt, bool more_tiles) {
    constexpr int CP = decltype(cp_tag)::value;
    if (!PDB) asm volatile("" : "+v"(zv));
    const int cc = c_begin + ci;
    const bool last_chunk = ci + 1 == ncl;
    const bool nextc = !last_chunk || more_tiles;             // a patch chunk follows this one
    const int coff_next = (last_chunk ? c_begin : cc + 1) * CK * 2;
    const int nbase = last_chunk ? nxt.base : cur.base;
    const unsigned nedge = last_chunk ? nxt.edge : cur.edge;
    const unsigned char* Acur = As + apar * A_BYTES;
    unsigned char* Anext = As + (apar ^ 1) * A_BYTES;
    const __amdgpu_buffer_rsrc_t rxn = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, nextc ? p.x_bytes : 0u, 0x00020000);
#pragma unroll
    for (int tg = 0; tg < NG; ++tg) {
      const int P = (CP + tg) & 1;                             // constant after unrolling
      const unsigned char* Bcur = Bs + P * B_BYTES;
      unsigned char* Bnext = Bs + (P ^ 1) * B_BYTES;
      // the stage two ahead: (chunk, tap group), wrapping into the next tile
      const int tg2 = (tg + 2) % NG;
      int ci2 = ci + (tg + 2) / NG;
      bool have2 = true;
      if (ci2 >= ncl) { ci2 -= ncl; have2 = more_tiles; }
      const int w2 = wbyte(c_begin + ci2, tg2);
      const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, have2 ? p.w_bytes : 0u, 0x00020000);
      // LDS byte offset of this lane's A fragment rows for one tap (k-step 0; k-step ks is that XOR (ks << 5))
      int aad0[TM], aad1[TM];
      auto tap_addr = [&](int tl, int (&aad)[TM]) {
        const int t = tg * TG + tl;
        int toff;
        if (KIND == KB_K3S1) { toff = (t / 3) * PW + (t % 3); }
        else if (KIND == KB_K4S2) { const int dy = t >> 2, dx = t & 3; toff = dy * PW + (dx & 1) * (PW >> 1) + (dx >> 1); }
        else { const int ta = t >> 1, tb2 = t & 1; toff = (py ? ta : 1 - ta) * PW + (px ? tb2 : 1 - tb2); }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int prow = arow[i] + zv + toff;
          aad[i] = prow * ROWB + ((lh ^ ((prow >> LGR) & (SEGS - 1))) << 4);
        }
      };
      // k-step stp reads tap stp / KS; consecutive k-steps alternate between the two address sets only when a tap is a
      // single k-step (never: KS >= 2), so one set per fragment register set suffices
      auto ldfr = [&](int stp, bf16x8 (&a)[TM], bf16x8 (&b)[TN], int (&aad)[TM]) {
        const int tl = stp / KS, ks = stp % KS;
        if constexpr (PADA) {
          const int t = tg * TG + tl, dy = t >> 2, dx = t & 3;
          const int cst = (dy * PWC + (dx & 1) * (PWC >> 1) + (dx >> 1)) * AROWB + ks * 32;   // constant after unrolling
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Acur + arow[i] + cst);
        } else {
          if (ks < 2) tap_addr(tl, aad);            // the first use of this tap by this register set
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(Acur + (aad[i] ^ (ks << 5)));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
          b[j] = *reinterpret_cast<const bf16x8*>(Bcur + tl * BN * ROWB + (boffs[j] ^ (ks << 5)));
      };
      auto mma = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      };
      // memory work of k-step slot k of this stage
      auto side = [&](int k) {
        if (k < HS) {
          // weights of stage + 2 into the set this stage's weights came from
#pragma unroll
          for (int q = 0; q < NBL; ++q)
            if (q * HS / NBL == k) rb[P][q] = bload16s(rw2, bgo0, w2 + bgo_s(q));
          if (tg < ALS) {
            const int g = tg * HS + k;
#pragma unroll
            for (int q = 0; q < NPL; ++q)
              if (q * AG / NPL == g) ra[q] = bload16s(rxn, avoff(q, nbase, nedge), coff_next);
          }
        } else {
          const int k2 = k - HS;
          // weights of stage + 1 (loaded during the previous stage) into the other LDS stage
#pragma unroll
          for (int q = 0; q < NBL; ++q)
            if (q * HS / NBL == k2) *reinterpret_cast<u32x4*>(Bnext + blo0 + q * BLO_STEP) = rb[P ^ 1][q];
          if (PDB && tg == NG - 1) {
#pragma unroll
            for (int q = 0; q < NPL; ++q)
              if (q * HS / NPL == k2) *adst(Anext, plo_of(q)) = ra[q];
          }
        }
      };
      bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
      ldfr(0, a0, b0, aad0);
#pragma unroll
      for (int s2 = 0; s2 < NS; s2 += 2) {
        ldfr(s2 + 1, a1, b1, aad1);
        side(s2);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (s2 + 2 < NS) ldfr(s2 + 2, a0, b0, aad0);
        side(s2 + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((DBG & 32) != 0) stamp(4 + 2 * min(ci * NG + tg, 22));
      __syncthreads();
      if constexpr ((DBG & 32) != 0) stamp(5 + 2 * min(ci * NG + tg, 22));
    }
    if (PDB) apar ^= 1;
  };

  auto chunk_boundary = [&](int ci) {
    if (!PDB && ci + 1 < ncl) {
      // single patch buffer: every wave is past its last read of this chunk
#pragma unroll
      for (int q = 0; q < NPL; ++q) *adst(As, plo_of(q)) = ra[q];
      __syncthreads();
    }
  };
  for (;;) {
    const int ntile = tile + (int)gridDim.x;
    const bool more_tiles = !PDB && ntile < ntiles;           // the two-patch-buffer forms are launched one block per tile
    const TilePos nxt = tile_pos(more_tiles ? ntile : tile);
    if constexpr ((NG & 1) == 0) {
      for (int ci = 0; ci < ncl; ++ci) {
        run_chunk(IC<0>{}, ci, nxt, more_tiles);
        chunk_boundary(ci);
      }
    } else {
      // odd number of stages per chunk: the weight register sets swap roles from one chunk to the next
      static_assert((NG & 1) == 0 || PDB, "the persistent form needs an even number of stages per tile");
      for (int ci = 0; ci < ncl; ci += 2) {
        run_chunk(IC<0>{}, ci, nxt, more_tiles);
        chunk_boundary(ci);
        if (ci + 1 < ncl) {
          run_chunk(IC<1>{}, ci + 1, nxt, more_tiles);
          chunk_boundary(ci + 1);
        }
      }
    }
    // every wave is past its last LDS read of this tile: the patch region is free for the epilogue's transpose.  The next
    // tile's first patch chunk is still in registers and its first weight stages sit behind the patch region.
    epilogue(cur, tile);
    if (!more_tiles) break;
    zero_acc();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NPL; ++q) *adst(As, plo_of(q)) = ra[q];
    __syncthreads();
    tile = ntile;
    cur = nxt;
  }
}

// ---- weights: packed fp32 P[Tsrc][R][C] -> bf16 Wb[phase][chunk][tap][Npad][CK] -------------------------------------
//   transpose = 0 (forward): n = column of P (cout), k = row of P (cin)
//   transpose = 1 (input gradient): n = row of P (cin), k = column of P (cout)
// One thread per 16-byte output segment; for transpose = 0 the 8 values of a segment are a column walk of P, so a
// 32 x 32 tile goes through LDS (reads coalesced along the columns of P, writes along k).
__device__ __forceinline__ int src_tap(int kind, int flip, int T, int t, int phase) {
  if (kind == KB_TCONV) {
    const int py = phase >> 1, px = phase & 1, a = t >> 1, b = t & 1;
    const int k4y = py ? (a ? 0 : 2) : (a ? 3 : 1);
    const int k4x = px ? (b ? 0 : 2) : (b ? 3 : 1);
    return k4y * 4 + k4x;
  }
  return flip ? (T - 1 - t) : t;
}

constexpr int PACK16_KT = 4;   // 32 x 32 tiles per block of the batched re-pack (even)

__device__ __forceinline__ void pack_bf16_block(const float* __restrict__ P, unsigned short* __restrict__ out, int R, int C,
                                                int kind, int flip, int transpose, int T, int Nn, int Npad, int Kk,
                                                int CK, int bx, int by, int zt, float (&tile)[32][33]) {
  const int phase = zt / T, t = zt - phase * T;   // zt = phase * T + t
  const int ts = src_tap(kind, flip, T, t, phase);
  const float* sp = P + (size_t)ts * R * C;
  const int n0 = bx * 32, k0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  // tile[kl][nl]
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = 0.f;
    if (!transpose) {
      const int k = k0 + ty + 8 * q, n = n0 + tx;      // P[k][n], coalesced along n
      if (k < Kk && n < Nn) v = sp[(size_t)k * C + n];
      tile[ty + 8 * q][tx] = v;
    } else {
      const int n = n0 + ty + 8 * q, k = k0 + tx;      // P[n][k], coalesced along k
      if (k < Kk && n < Nn) v = sp[(size_t)n * C + k];
      tile[tx][ty + 8 * q] = v;
    }
  }
  __syncthreads();
  // 32 n x 4 segments of 8 k
  if (threadIdx.x < 128) {
    const int nl = threadIdx.x >> 2, sg = threadIdx.x & 3;
    const int n = n0 + nl, k = k0 + sg * 8;
    if (n < Npad && k < Kk) {
      const int nchunk = Kk / CK;
      const int cc = k / CK, kc = k - cc * CK;
      u32x4 v;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        f32x2 f = {tile[sg * 8 + 2 * h][nl], tile[sg * 8 + 2 * h + 1][nl]};
        v[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
      }
      const size_t o = ((((size_t)phase * nchunk + cc) * T + t) * Npad + n) * CK + kc;
      *reinterpret_cast<u32x4*>(out + o) = v;
    }
  }
}

// PACK16_KT tiles of 32 x 32 in one block, side by side along the SOURCE-contiguous dimension (n for the forward layout, k for
// the transposed one): all loads of the strip are issued before the one barrier (16 in flight per thread instead of 4), and a
// source row contributes 512 contiguous bytes to a block instead of 128 (round 3: D_NET256's re-pack 283 -> 237 us with the
// tiles walked one after another, -> see profiles/README.md for this form).
__device__ __forceinline__ void pack_bf16_strip(const s2i_pack16_item& it, int bx, int by, int zt,
                                                float (&tile)[PACK16_KT][33][33]) {   // 33 x 33: tile i starts one bank later
  const int T = it.T, C = it.C, Kk = it.Kk, Nn = it.Nn, Npad = it.Npad, CK = it.CK;
  const int phase = zt / T, t = zt - phase * T;
  const int ts = src_tap(it.kind, it.flip, T, t, phase);
  const float* sp = it.P + (size_t)ts * it.R * C;
  // 16-byte loads: a strip row is PACK16_KT * 32 floats = 32 lanes x 4; eight rows per pass, four passes
  static_assert(PACK16_KT == 4, "a strip row is covered by 32 lanes of four floats");
  const int c4 = threadIdx.x & 31, r8 = threadIdx.x >> 5;
  const int i = c4 >> 3, cl = (c4 & 7) * 4;          // tile of the strip, first of the lane's four columns inside it
  const bool vec_ok = (C & 3) == 0 && ((size_t)sp & 15) == 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int rl = r8 + 8 * q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!it.transpose) {
      const int k = by * 32 + rl, n = (bx * PACK16_KT) * 32 + c4 * 4;      // P[k][n .. n + 3]
      if (k < Kk && n < Nn) {
        const float* src = sp + (size_t)k * C + n;
        if (vec_ok && n + 3 < C) v = *reinterpret_cast<const f32x4*>(src);
        else
#pragma unroll
          for (int j = 0; j < 4; ++j) if (n + j < C) v[j] = src[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (n + j >= Nn) v[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[i][rl][cl + j] = v[j];
    } else {
      const int n = bx * 32 + rl, k = (by * PACK16_KT) * 32 + c4 * 4;      // P[n][k .. k + 3]
      if (n < Nn && k < Kk) {
        const float* src = sp + (size_t)n * C + k;
        if (vec_ok && k + 3 < C) v = *reinterpret_cast<const f32x4*>(src);
        else
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < C) v[j] = src[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (k + j >= Kk) v[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) tile[i][cl + j][rl] = v[j];
    }
  }
  __syncthreads();
  const int half = threadIdx.x >> 7, w = threadIdx.x & 127;
  const int nl = w >> 2, sg = w & 3;
  const int nchunk = Kk / CK;
#pragma unroll
  for (int j = 0; j < PACK16_KT / 2; ++j) {
    const int i = 2 * j + half;
    const int n0 = (it.transpose ? bx : bx * PACK16_KT + i) * 32, k0 = (it.transpose ? by * PACK16_KT + i : by) * 32;
    const int n = n0 + nl, k = k0 + sg * 8;
    if (n < Npad && k < Kk) {
      const int cc = k / CK, kc = k - cc * CK;
      u32x4 v;
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        f32x2 f = {tile[i][sg * 8 + 2 * h][nl], tile[i][sg * 8 + 2 * h + 1][nl]};
        v[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
      }
      const size_t o = ((((size_t)phase * nchunk + cc) * T + t) * Npad + n) * CK + kc;
      *reinterpret_cast<u32x4*>(it.out + o) = v;
    }
  }
}

__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ P, unsigned short* __restrict__ out, int R,
                                                        int C, int kind, int flip, int transpose, int T, int nphase,
                                                        int Nn, int Npad, int Kk, int CK) {
  __shared__ float tile[32][33];
  pack_bf16_block(P, out, R, C, kind, flip, transpose, T, Nn, Npad, Kk, CK, blockIdx.x, blockIdx.y, blockIdx.z, tile);
}

// every bf16 weight copy of a network in ONE launch (after the fused Adam): item k owns the linear blocks
// [block0, block0 + gx * gy * nphase * T), block0 ascending
__global__ __launch_bounds__(256) void pack_bf16_batched_kernel(const s2i_pack16_item* __restrict__ items, int n) {
  __shared__ float tile[PACK16_KT][33][33];
  const int b = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {                       // last item with block0 <= b
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block0 <= b) lo = mid;
    else hi = mid - 1;
  }
  const s2i_pack16_item it = items[lo];
  const int r = b - it.block0;
  const int bx = r % it.gx, by = (r / it.gx) % it.gy, zt = r / (it.gx * it.gy);   // it.gx / it.gy count STRIPS
  pack_bf16_strip(it, bx, by, zt, tile);
}

// ---- split-K reduction with bf16 output (+ BatchNorm column statistics), as splitk_reduce_stats_kernel -----------------
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ slab, int S, long long rows,
                                                                 int N, unsigned short* __restrict__ y, int ldy,
                                                                 float* __restrict__ part, int nparts, int cpb, int ppg,
                                                                 long long Rg) {
  __shared__ f32x4 sh[2][256];
  const int tid = threadIdx.x;
  const int rpb = 256 / cpb;
  const int ql = tid % cpb, rl = tid / cpb;
  const int quad = blockIdx.y * cpb + ql;
  const int Q = N / 4;
  const int grp = blockIdx.x / ppg, pp = blockIdx.x - grp * ppg;
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
      u32x2 o = {pack2(v[0], v[1]), pack2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(y + row * ldy + quad * 4) = o;
      s0 += v;
      s1 += v * v;
    }
  }
  if (!part) return;
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

// ---- host-side planning ------------------------------------------------------------------------------------------
struct BPlan {
  int v2;   // conv_bf16_v2_kernel: 256-pixel tiles, 512 threads
  int kb, T, NG, TG, Ho, Wo, nphases, BN, CK, Npad;
  int lgTW, lgTH, lgTB, tilesX, tilesY, tilesB, PH, PW, npix;
  int nchunk, splitk, cps, gridM, gridN;
  long long Mrows;
};

int plan_bf16(const s2i_conv_desc* d, BPlan* pl) {
  S2I_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->N > 0, "conv(bf16): non-positive extent");
  S2I_REQUIRE(d->Cc == 0, "conv(bf16): broadcast vectors are concatenated by the caller");
  S2I_REQUIRE(s2i_is_pow2(d->H) && s2i_is_pow2(d->W), "conv(bf16): spatial extents must be powers of two");
  S2I_REQUIRE(d->Cx >= 32 && (d->Cx % 32) == 0, "conv(bf16): input channels must be a multiple of 32 (Cx=%d)", d->Cx);
  S2I_REQUIRE((d->N % 8) == 0 && (d->ldy % 8) == 0 && d->ldy >= d->N, "conv(bf16): N and ldy must be multiples of 8");
  S2I_REQUIRE(d->act == S2I_ACT_NONE, "conv(bf16): no activation epilogue");
  pl->nphases = 1;
  switch (d->kind) {
    case S2I_CONV_K3S1: pl->kb = KB_K3S1; pl->T = 9; pl->NG = 1; pl->Ho = d->H; pl->Wo = d->W; break;
    case S2I_CONV_K4S2:
      S2I_REQUIRE(d->H >= 2 && d->W >= 2, "conv(bf16) k4s2: extent < 2");
      pl->kb = KB_K4S2; pl->T = 16; pl->NG = 2; pl->Ho = d->H / 2; pl->Wo = d->W / 2; break;
    case S2I_TCONV_K4S2: pl->kb = KB_TCONV; pl->T = 4; pl->NG = 1; pl->Ho = d->H; pl->Wo = d->W; pl->nphases = 4; break;
    default: S2I_FAIL("conv(bf16): unsupported kind %d", d->kind);
  }
  // Stage shape: all 9 taps of a 3x3 / the 4 taps of a transposed-conv phase per stage; the stride-2 4x4 conv with BN = 128
  // takes 4 taps x 32 channels (64-byte pieces of a pixel instead of 32-byte ones: measured -10 % time, the 32-byte pieces
  // left half of every 64-byte memory request unused).
  pl->BN = d->N > 64 ? 128 : (d->N > 32 ? 64 : 32);
  pl->TG = pl->kb == KB_K4S2 ? 8 : pl->T;
  const bool wide_ck = pl->BN == 128 && pl->kb == KB_K4S2;
  if (wide_ck) pl->TG = 4;
  pl->NG = pl->T / pl->TG;
  int ck = (pl->kb == KB_TCONV ? 4096 : 2048) / pl->BN;
  if (ck > (pl->kb == KB_K4S2 ? 32 : 64)) ck = pl->kb == KB_K4S2 ? 32 : 64;  // the stride-2 patch is 5 pixels per output pixel
  if (wide_ck) ck = 32;
  while (ck > 16 && (d->Cx % ck) != 0) ck >>= 1;
  if (pl->BN == 32 && ck < 32) ck = 32;
  S2I_REQUIRE((d->Cx % ck) == 0, "conv(bf16): %d channels do not split into chunks of %d", d->Cx, ck);
  pl->CK = ck;
  pl->Npad = s2i_cdiv(d->N, pl->BN) * pl->BN;
  // second-generation kernel (256-pixel tiles); tuning knob b16_v2 = 0 never, 1 (default) where it was measured faster, 2
  // wherever it can run (tests)
  const int v2mode = s2i_tune(S2I_TUNE_B16_V2, 1);
  pl->v2 = 0;
  if (v2mode && pl->BN == 128) {
    const int ck2 = pl->kb == KB_TCONV ? 64 : 32;
    int tw2 = pl->Wo < 32 ? pl->Wo : 32, th2 = 256 / tw2;
    if (th2 > pl->Ho) th2 = pl->Ho;
    const int tb2 = 256 / (tw2 * th2);
    int ph2, pw2;
    if (pl->kb == KB_K3S1) { ph2 = th2 + 2; pw2 = tw2 + 2; }
    else if (pl->kb == KB_K4S2) { ph2 = 2 * th2 + 2; pw2 = 2 * tw2 + 2; }
    else { ph2 = th2 + 1; pw2 = tw2 + 1; }
    const int npix2 = tb2 * ph2 * pw2;
    const int maxpix2 = pl->kb == KB_K4S2 ? 1280 : (pl->kb == KB_K3S1 ? 416 : 336);
    bool ok = (d->Cx % ck2) == 0 && npix2 <= maxpix2;
    if (d->stats && d->groups > 1) ok = ok && pl->kb != KB_TCONV && (d->B % d->groups) == 0 && ((d->B / d->groups) % tb2) == 0;
    const long long blocks2 = (long long)(pl->Wo / tw2) * (pl->Ho / th2) * s2i_cdiv(d->B, tb2) * (pl->Npad / 128) * pl->nphases;
    if (v2mode == 1) ok = ok && blocks2 >= 224;
    if (ok) {
      pl->v2 = 1;
      pl->CK = ck = ck2;
      pl->TG = pl->kb == KB_K4S2 ? 4 : (pl->kb == KB_K3S1 ? 3 : 2);
      pl->NG = pl->T / pl->TG;
    }
  }
  const int bm = pl->v2 ? 256 : 128;
  // tile of 128 (256) output pixels: as wide as the map allows (up to 32), then rows, then images
  int tw = pl->Wo < 32 ? pl->Wo : 32;
  int th = bm / tw;
  if (th > pl->Ho) th = pl->Ho;
  int tb = bm / (tw * th);
  pl->lgTW = s2i_ilog2(tw); pl->lgTH = s2i_ilog2(th); pl->lgTB = s2i_ilog2(tb);
  pl->tilesX = pl->Wo / tw; pl->tilesY = pl->Ho / th; pl->tilesB = s2i_cdiv(d->B, tb);
  if (pl->kb == KB_K3S1) { pl->PH = th + 2; pl->PW = tw + 2; }
  else if (pl->kb == KB_K4S2) { pl->PH = 2 * th + 2; pl->PW = 2 * tw + 2; }
  else { pl->PH = th + 1; pl->PW = tw + 1; }
  pl->npix = tb * pl->PH * pl->PW;
  if (!pl->v2 && wide_ck && pl->npix > 672 && ck == 32) { pl->CK = ck = 16; pl->TG = 8; pl->NG = 2; }   // 8 maps of 4x4 outputs per tile
  S2I_REQUIRE(pl->v2 || pl->npix <= (pl->kb == KB_K4S2 ? (ck == 16 ? 800 : 672) : 320),
              "conv(bf16): patch of %d pixels exceeds the LDS plan", pl->npix);
  pl->gridM = pl->tilesX * pl->tilesY * pl->tilesB;
  pl->gridN = pl->Npad / pl->BN;
  pl->nchunk = d->Cx / ck;
  const long long M = (long long)d->B * pl->Ho * pl->Wo;
  pl->Mrows = M * pl->nphases;
  S2I_REQUIRE(pl->Mrows * d->ldy < (1ll << 31), "conv(bf16): output too large");
  if (d->stats && d->groups > 1) {
    S2I_REQUIRE(pl->kb != KB_TCONV && (d->B % d->groups) == 0 && ((d->B / d->groups) % tb) == 0,
                "conv(bf16): a tile of %d images straddles the %d BatchNorm groups of batch %d", tb, d->groups, d->B);
  }
  const long long blocks = (long long)pl->gridM * pl->gridN * pl->nphases;
  int splitk = 1;
  const int slots = pl->v2 ? 256 : 512;                 // resident blocks of the chip
  if (blocks < slots * 3 / 4 && pl->nchunk >= 4 && !d->nosplit) {
    splitk = (int)(slots / blocks);
    if (splitk > pl->nchunk / 2) splitk = pl->nchunk / 2;
    if (splitk > 32) splitk = 32;
    if (splitk < 1) splitk = 1;
  }
  if (!pl->v2 && pl->nchunk >= 4 && !d->nosplit && blocks < 4 * slots) {
    // 128-pixel kernel, a launch of a few blocks per CU: the time is set by the CU that holds one block more than the others
    // (the discriminators' 8 -> 4 px layer at batch 48: 288 blocks on 256 CUs ran like 512).  Price every split by the work
    // of the most loaded CU plus the slab traffic it adds, in us: a CU does ~3.9 TFLOP/s with two resident blocks (0.6 of its
    // clocked bf16 peak, profiles/r03_bf16_conv_layers.txt), ~0.75 of that with one; slabs move at ~4 TB/s; the reduction is a
    // launch of its own (~12 us in its chain).
    const double tile_us = 2.0 * 128.0 * pl->BN * (double)pl->T * d->Cx / 3.9e6;
    const double slab_us = (double)pl->Mrows * d->N * 8.0 / 4.0e6;
    double best = 1e300;
    int best_s = splitk;
    const int smax = pl->nchunk / 2 < 32 ? pl->nchunk / 2 : 32;
    for (int sc = 1; sc <= smax; ++sc) {
      const int cps = s2i_cdiv(pl->nchunk, sc), se = s2i_cdiv(pl->nchunk, cps);
      if (se != sc) continue;
      const long long nb = blocks * se;
      const double per_cu = (double)((nb + 255) / 256);
      double cost = per_cu / se * tile_us * (nb <= 256 ? 1.0 / 0.75 : 1.0) + 2.5 * per_cu;   // + a block's prologue / epilogue
      if (se > 1) cost += se * slab_us + 12.0;
      if (cost < best) { best = cost; best_s = se; }
    }
    splitk = best_s;
  }
  pl->cps = s2i_cdiv(pl->nchunk, splitk);
  pl->splitk = s2i_cdiv(pl->nchunk, pl->cps);
  return 0;
}

int bf16_stat_parts(const BPlan& pl, int groups) {
  if (groups < 1) groups = 1;
  if (pl.splitk > 1) {
    int ppg = s2i_cdiv(pl.Mrows / groups, 8);
    if (ppg > 512 / groups) ppg = 512 / groups;
    if (ppg < 1) ppg = 1;
    return ppg * groups;
  }
  return pl.gridM * pl.nphases;
}

// the 66-pixel-wide stride-2 patch (32-column tiles) takes the padded-row instantiation
bool v2_padded(const BPlan& pl) { return pl.v2 && pl.kb == KB_K4S2 && pl.PW == 66 && pl.CK == 32; }

size_t bf16_smem_bytes(const BPlan& pl) {
  const int rowb = pl.CK * 2, tg = pl.TG;
  size_t ab = ((size_t)pl.npix * rowb + 255) & ~(size_t)255;
  if (pl.v2) {
    if (v2_padded(pl)) ab = ((size_t)pl.npix * (rowb + 16) + 255) & ~(size_t)255;
    const size_t main2 = (pl.kb == KB_K4S2 ? 1 : 2) * ab + 2 * (size_t)tg * 128 * rowb + 1024;
    const size_t epi2 = (size_t)256 * (128 * 2 + 16);
    return main2 > epi2 ? main2 : epi2;
  }
  const size_t main_b = ab + (size_t)tg * pl.BN * rowb;
  const size_t epi = (size_t)128 * (pl.BN * 2 + 16);
  return main_b > epi ? main_b : epi;
}

bool bf16_has_kernel(int kb, int bn, int ck) {
  if (kb == KB_TCONV) return (bn == 128 && ck == 32) || (bn == 64 && ck == 64) || (bn == 32 && (ck == 64 || ck == 32));
  if (kb == KB_K4S2) return (bn == 128 && (ck == 16 || ck == 32)) || (bn == 64 && ck == 32) || (bn == 32 && ck == 32);
  return (bn == 128 && (ck == 16 || ck == 32)) || (bn == 64 && ck == 32) || (bn == 32 && (ck == 64 || ck == 32));
}

template <int KIND, int BN, int CK, int TG>
int launch_one(const BPlan& pl, const ConvBP& p, dim3 grid, hipStream_t st) {
  const size_t shb = bf16_smem_bytes(pl);
  static bool raised = false;  // > 64 KB of dynamic LDS needs the attribute once per kernel
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_bf16_kernel<KIND, BN, CK, TG>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) S2I_FAIL("conv(bf16): hipFuncSetAttribute: %s", hipGetErrorString(e));
    raised = true;
  }
  hipLaunchKernelGGL((conv_bf16_kernel<KIND, BN, CK, TG>), grid, dim3(256), shb, st, p);
  return 0;
}

// blocks along x for the second-generation kernel: persistent (one block per CU looping over its tiles) where the kernel
// supports it -- single patch buffer, no split-K, and a patch region large enough for the epilogue's transpose
int v2_grid_x(const BPlan& pl) {
  const int pers = s2i_tune(S2I_TUNE_B16_PERSIST, 1);
  const size_t ab = ((size_t)pl.npix * (pl.CK * 2 + (v2_padded(pl) ? 16 : 0)) + 255) & ~(size_t)255;
  if (!pers || pl.kb != KB_K4S2 || pl.splitk != 1 || ab < (size_t)256 * (128 * 2 + 16)) return pl.gridM;
  int nblk = (pers > 1 ? pers : 256) / (pl.gridN * pl.nphases);   // knob b16_persist > 1: that many block slots (tests)
  if (nblk < 1) nblk = 1;
  if (pl.gridM <= nblk) return pl.gridM;
  // a block walks ceil(gridM / nblk) tiles: persistent only where that rounding costs little (measured: 144 tiles over
  // 64 slots -- 3 rounds for 2.25 rounds of work -- lost what the prefetch across tiles gained)
  if (pers == 1 && (pl.gridM % nblk) != 0 && pl.gridM < 6 * nblk) return pl.gridM;
  return nblk;
}

template <int KIND, int CK, int TG, bool PDB, int PWC = 0>
int launch_v2(const BPlan& pl, const ConvBP& p, dim3 grid, hipStream_t st) {
  const size_t shb = bf16_smem_bytes(pl);
  S2I_REQUIRE(shb <= 160 * 1024, "conv(bf16): %zu bytes of LDS", shb);
  grid.x = v2_grid_x(pl);
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_bf16_v2_kernel<KIND, CK, TG, PDB, 0, PWC>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) S2I_FAIL("conv(bf16): hipFuncSetAttribute: %s", hipGetErrorString(e));
    raised = true;
  }
  hipLaunchKernelGGL((conv_bf16_v2_kernel<KIND, CK, TG, PDB, 0, PWC>), grid, dim3(512), shb, st, p);
  return 0;
}

#ifdef S2I_DIAG
#include "diag/s2i_bf16_diag.inc"
#endif

int launch_conv_bf16(const BPlan& pl, const ConvBP& p, dim3 grid, hipStream_t st) {
  const int kb = pl.kb, bn = pl.BN, ck = pl.CK, tg = pl.TG;
#ifdef S2I_DIAG
  {   // libs2i_hip_diag.so only (make diag): ablation / in-kernel timeline instantiations of tools/conv16_*.py
    int rc = 0;
    if (diag_launch_conv_bf16(pl, p, grid, st, &rc)) return rc;
  }
#endif
  if (pl.v2) {
    if (kb == KB_K4S2 && v2_padded(pl)) return launch_v2<KB_K4S2, 32, 4, false, 66>(pl, p, grid, st);
    if (kb == KB_K4S2) return launch_v2<KB_K4S2, 32, 4, false>(pl, p, grid, st);
    if (kb == KB_K3S1) return launch_v2<KB_K3S1, 32, 3, true>(pl, p, grid, st);
    return launch_v2<KB_TCONV, 64, 2, true>(pl, p, grid, st);
  }
#define S2I_CASE(K, bn_, ck_, tg_) \
  if (kb == K && bn == bn_ && ck == ck_ && tg == tg_) return launch_one<K, bn_, ck_, tg_>(pl, p, grid, st);
  S2I_CASE(KB_K3S1, 128, 16, 9) S2I_CASE(KB_K3S1, 64, 32, 9) S2I_CASE(KB_K3S1, 32, 64, 9) S2I_CASE(KB_K3S1, 32, 32, 9)
  S2I_CASE(KB_K4S2, 128, 32, 4) S2I_CASE(KB_K4S2, 128, 16, 8) S2I_CASE(KB_K4S2, 64, 32, 8) S2I_CASE(KB_K4S2, 32, 32, 8)
  S2I_CASE(KB_TCONV, 128, 32, 4) S2I_CASE(KB_TCONV, 64, 64, 4) S2I_CASE(KB_TCONV, 32, 64, 4) S2I_CASE(KB_TCONV, 32, 32, 4)
#undef S2I_CASE
  S2I_FAIL("conv(bf16): no kernel for kind %d BN=%d CK=%d TG=%d", kb, bn, ck, tg);
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int s2i_conv_bf16_eligible(const s2i_conv_desc* d) {
  BPlan pl;
  const int rc = plan_bf16(d, &pl);
  if (rc) return 0;
  return (pl.v2 || bf16_has_kernel(pl.kb, pl.BN, pl.CK)) ? 1 : 0;
}

// identifies the arrangement of the bf16 weights the plan of this descriptor consumes (a cache key for callers: the same
// layer at another batch size may be planned onto another kernel)
extern "C" int s2i_conv_bf16_weight_layout(const s2i_conv_desc* d) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return -1;
  return pl.CK | (pl.Npad << 8);
}

extern "C" size_t s2i_conv_bf16_workspace_bytes(const s2i_conv_desc* d) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return 0;
  return pl.splitk > 1 ? (size_t)pl.splitk * pl.Mrows * d->N * sizeof(float) : 0;
}

extern "C" int s2i_conv_bf16_stat_parts(const s2i_conv_desc* d) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return -1;
  return bf16_stat_parts(pl, d->groups);
}

extern "C" size_t s2i_conv_bf16_weight_elems(const s2i_conv_desc* d) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return 0;
  return (size_t)pl.nphases * pl.T * pl.Npad * d->Cx;
}

extern "C" int s2i_pack_conv_weight_bf16(const s2i_conv_desc* d, const float* packed, int R, int C, unsigned short* out,
                                         void* stream) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return 1;
  S2I_REQUIRE(packed && out, "pack(bf16): null pointer");
  const int transpose = d->wmode != 0;
  const int Kk = d->Cx, Nn = d->N;
  if (transpose) S2I_REQUIRE(R >= Nn && C >= Kk, "pack(bf16): P is %d x %d, need rows >= %d cols >= %d", R, C, Nn, Kk);
  else S2I_REQUIRE(R >= Kk && C >= Nn, "pack(bf16): P is %d x %d, need rows >= %d cols >= %d", R, C, Kk, Nn);
  dim3 grid(pl.Npad / 32, Kk / 32, pl.nphases * pl.T);
  hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, ST, packed, out, R, C, pl.kb, d->flip, transpose, pl.T,
                     pl.nphases, Nn, pl.Npad, Kk, pl.CK);
  S2I_LAUNCH_CHECK("pack_bf16");
  return 0;
}

extern "C" int s2i_pack16_item_fill(const s2i_conv_desc* d, const float* packed, int R, int C, unsigned short* out,
                                    s2i_pack16_item* item) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return -1;
  S2I_REQUIRE(packed && out && item, "pack(bf16): null pointer");
  const int transpose = d->wmode != 0;
  const int Kk = d->Cx, Nn = d->N;
  if (transpose) S2I_REQUIRE(R >= Nn && C >= Kk, "pack(bf16): P is %d x %d, need rows >= %d cols >= %d", R, C, Nn, Kk);
  else S2I_REQUIRE(R >= Kk && C >= Nn, "pack(bf16): P is %d x %d, need rows >= %d cols >= %d", R, C, Kk, Nn);
  item->P = packed; item->out = out; item->R = R; item->C = C; item->kind = pl.kb; item->flip = d->flip;
  item->transpose = transpose; item->T = pl.T; item->nphase = pl.nphases; item->Nn = Nn; item->Npad = pl.Npad;
  item->Kk = Kk; item->CK = pl.CK; item->block0 = 0;
  // strips of PACK16_KT tiles along the source-contiguous dimension (pack_bf16_strip)
  item->gx = transpose ? pl.Npad / 32 : (pl.Npad / 32 + PACK16_KT - 1) / PACK16_KT;
  item->gy = transpose ? (Kk / 32 + PACK16_KT - 1) / PACK16_KT : Kk / 32;
  return item->gx * item->gy * pl.nphases * pl.T;   // blocks of this item
}

extern "C" int s2i_pack_conv_weights_bf16_batched(const s2i_pack16_item* items_dev, int n, int total_blocks, void* stream) {
  S2I_REQUIRE(items_dev && n > 0 && total_blocks > 0, "pack(bf16, batched): empty table");
  hipLaunchKernelGGL(pack_bf16_batched_kernel, dim3(total_blocks), dim3(256), 0, ST, items_dev, n);
  S2I_LAUNCH_CHECK("pack_bf16_batched");
  return 0;
}

extern "C" int s2i_conv_forward_bf16(const s2i_conv_desc* d, const unsigned short* x, const unsigned short* w,
                                     const float* cls_bias, unsigned short* y, float* part, void* ws, size_t ws_bytes,
                                     void* stream) {
  BPlan pl;
  if (plan_bf16(d, &pl)) return 1;
  S2I_REQUIRE(x && w && y, "conv(bf16): null operand");
  S2I_REQUIRE(!cls_bias || (d->kind == S2I_CONV_K3S1 && pl.splitk == 1), "conv(bf16): class bias needs an unsplit 3x3 conv");
  S2I_REQUIRE(!d->stats || part, "conv(bf16): stats requested without a partial buffer");
  const size_t need = pl.splitk > 1 ? (size_t)pl.splitk * pl.Mrows * d->N * sizeof(float) : 0;
  S2I_REQUIRE(ws_bytes >= need && (need == 0 || ws), "conv(bf16): workspace too small (%zu < %zu)", ws_bytes, need);
  ConvBP p;
  p.x = x; p.w = w; p.cls_bias = cls_bias; p.y = y; p.part = part; p.slab = (float*)ws;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C = d->Cx; p.Ho = pl.Ho; p.Wo = pl.Wo;
  p.N = d->N; p.Npad = pl.Npad; p.ldy = d->ldy;
  p.lgTW = pl.lgTW; p.lgTH = pl.lgTH; p.lgTB = pl.lgTB; p.tilesX = pl.tilesX; p.tilesY = pl.tilesY;
  p.ntiles = pl.gridM;
  p.PH = pl.PH; p.PW = pl.PW; p.npix = pl.npix;
  p.nchunk = pl.nchunk; p.splitk = pl.splitk; p.cps = pl.cps;
  p.stats = d->stats; p.nparts = pl.gridM * pl.nphases; p.Mrows = pl.Mrows;
  const unsigned long long xb = (unsigned long long)d->B * d->H * d->W * d->Cx * 2ull;
  const unsigned long long wb = (unsigned long long)pl.nphases * pl.T * pl.Npad * d->Cx * 2ull;
  S2I_REQUIRE(xb < 0x7ff00000ull && wb < 0x7ff00000ull, "conv(bf16): tensor exceeds the 2 GiB buffer-addressing window");
  p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
#ifdef S2I_DIAG
  p.dbg = s2i_tune(S2I_TUNE_B16_DBG, 0);
#else
  p.dbg = 0;
#endif
  dim3 grid(pl.gridM, pl.gridN, pl.nphases * pl.splitk);
  if (launch_conv_bf16(pl, p, grid, ST)) return 1;
  S2I_LAUNCH_CHECK("conv_bf16");
  if (pl.splitk > 1) {
    S2I_REQUIRE((d->N % 4) == 0, "conv(bf16): split-K needs N %% 4 == 0");
    const int Q = d->N / 4;
    int cpb = 1;
    while (cpb < Q && cpb < 256) cpb <<= 1;
    const int groups = d->groups < 1 ? 1 : d->groups;
    const int nparts = bf16_stat_parts(pl, groups);
    hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(nparts, (Q + cpb - 1) / cpb), dim3(256), 0, ST, (const float*)ws,
                       pl.splitk, pl.Mrows, d->N, y, d->ldy, d->stats ? part : nullptr, nparts, cpb, nparts / groups,
                       pl.Mrows / groups);
    S2I_LAUNCH_CHECK("splitk_reduce_bf16");
  }
  return 0;
}
