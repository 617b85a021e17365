// dc_mfma_kernels.hpp -- matrix-core variants of the two pairwise sweeps (gfx950,
// v_mfma_f32_32x32x16_f16 on fp32 coordinates split into two fp16 pieces).  Included by
// dc_mfma.hip (host side) and dc_mfma_step.hip (one translation unit per MFMA count).
//
// Idea.  The N x D . D x N distance block is a dense contraction:
//     d2(x, y) = |x|^2 + |y|^2 - 2 x.y
// so a 32(reference) x 32(query) tile is a few MFMAs instead of 32*32*(3D-1) VALU ops.
// But the Gram form does not round like the reference's direct-difference sum, and the outputs
// (integer populations, neighbour indices) must equal the reference's bit for bit.  So the MFMA
// result is used only as a CLASSIFIER with a rigorous guard band eps (derivation in DESIGN.md):
//     acc <  thr - eps   =>  canonical d2 <  thr      (decided by the MFMA value alone)
//     acc >= thr + eps   =>  canonical d2 >= thr
//     otherwise          =>  the pair is re-evaluated in the canonical order (dist2_canon_rt) from
//                            the ORIGINAL coordinates, in-kernel, by the lane that owns it.
// A handful of pairs per frame fall in the band (|d2 - r^2| < ~1e-5), so the re-check costs a few %.
// The nearest-neighbour sweep uses the same band around the running minimum: every reference frame
// whose MFMA distance is within 2.5 eps of the running minimum is evaluated exactly and merged
// lexicographically on (d2, index), which reproduces "lowest index wins ties" (:270) exactly.
//
// Why fp16 pieces.  On gfx950 the fp32-input MFMA runs at the vector rate and shares the vector
// pipe (its 64 cycles per K=2 step do not overlap the VALU epilogue: measured); the
// 16-bit MFMAs are 16x faster per flop and run beside the VALU.  The accumulation of any MFMA chain
// already costs tens of u = 2^-24 relative to max |x'|^2 (every addend is truncated to 2^-24 of the
// largest one), so the operands need to carry the coordinates to about 2^-22, not to the last bit:
// two fp16 pieces do that, x' = hi + mid + rho with |rho| <= 2^-22 |x'|, and of the products of
// x_k * y_k only the three largest are kept:
//     -2 x.y  ~  sum_k  yh*xh + ym*xh + yh*xm                               (x pieces of -2x')
// (the dropped ym*xm, yh*rho, rho*xh are <= 3 * 2^-22 |x_k y_k|: 27 u max|x'|^2 in the band, against
// 34 u for the accumulation itself).  That is 3 D "slots" of the K axis, plus two slots
// (2^a * pieces of c_q / 2^a, c_q a per-query constant) that fold the query norm and the threshold
// into the accumulator:
//     acc = |y'|^2 + c_q - 2 x'.y'   with  c_q = |x'|^2 - (r^2 - 1)     (populations: inside <=> acc < 0)
//                                          c_q = |x'|^2                 (neighbours:  acc ~ d2)
// NM = ceil((3 D + 2) / 16) MFMAs per tile: 2 for D = 10 (the first, exact bf16x3 version of these
// kernels -- git tag bf16x3-r1 -- needed 4; the fp32 MFMA 5 of 4x the cycles), 7 for D = 32, 13 for D = 64 (kMaxCols).
// fp16 has a narrow exponent range, so everything the matrix pipe sees is SCALED by powers of two
// chosen per SWEEP (exact; "scale of a SWEEP" below): the neighbour sweeps put S max|x'|^2 into
// [2^26, 2^28) (|x''_k| < 2^14, |-2 x''_k| < 2^15, |c_q| / 2^15 < 65504), the population sweeps take
// the largest S at which the guard band is <= 1, which turns the band test into a test of bit 30.
// Values below the smallest normal fp16 (2^-14) are stored as zero by the image
// builder (subnormal MFMA inputs are not exact on this hardware: scratch/mfma_probe_f16.hip); the
// flush is part of the band.  Norms, thresholds and bands live in the same scaled units; the exact
// path works on the original coordinates and never sees the scale.
// Slot order: the constant and the hi*hi products come FIRST, so after the first MFMA(s) the
// accumulator of a pair near the threshold is already small and the remaining (small) terms are
// added at a small magnitude -- that keeps the accumulation error, and with it eps, at the level of
// a single MFMA (the hardware truncates every addend to 2^-24 of the largest one and rounds once
// per MFMA: measured by scratch/mfma_probe_f16.hip, asserted by tests/cpp/test_mfma_model).
//
// Mapping (one wave = TQ query tiles of 32 frames, swept against reference tiles of 32 frames):
//   v_mfma_f32_32x32x16_f16:   D[i][j] = C[i][j] + sum_k A[i][k] B[k][j];  lane l = (r = l&31, h = l>>5)
//   holds A[r][8h..8h+7] and B[8h..8h+7][r] (8 fp16 = 4 VGPRs each); D[i][j] sits in lane (j + 32*h),
//   register g with i = (g&3) + 8*(g>>2) + 4*h.
//     A = reference pieces (streamed: NM x 16 B per lane and tile)
//     B = query pieces of -2x' and of c_q (resident in VGPRs for the whole sweep)
//     C = |y'_i|^2 broadcast along the row (initial accumulator, 4 x dwordx4; +inf for pad rows)
// Queries sit on the lane axis, so populations / running minima are per-lane registers; the two
// half-waves (h = 0/1) see disjoint reference rows of the same 32 queries and are merged by one
// __shfl_xor(.., 32) at the very end.
//
// The epilogues use no compare masks at all (no SGPR hand-offs between VALU and SALU).  Populations
// (threshold folded as c_q = |x'|^2 - (r^2 - 1), band <= 1):
//     inside  <=>  sign bit of acc             -\  both shifted into a bit string (one v_alignbit per
//     outside <=>  bit 30 (acc >= 2)           -/  element), v_bcnt of the sign positions
//     in band <=>  neither                     -> one v_bitop3 + compare per chain
// which is valid because the MFMA kernels only run on finite data (the header pass raises a flag
// for non-finite or overflow-prone rows; the flagged case runs the direct kernels instead, both
// launches are gated on the device so no host synchronisation is needed).
//
// Operand images (built per call in the caller's workspace, dc_mfma.hip):
//   A form [T][NM][64] x 16 B: fragment of lane l of MFMA m of tile t = slots 16m + 8(l>>5) + 0..7 of
//                              row 32t + (l&31); B form: the same with the query-side pieces
//   norms  [32T]               |y'|^2 (double accumulate, rounded once); +inf for the pad rows
#pragma once
#include "dc_mfma.hpp"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

namespace dc {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxCols = 64;           // n_cols handled by the matrix-core kernels
constexpr int kConstSlots = 2;         // K slots 0..1: 2^15 (A side) x pieces of c_q / 2^15 (B side)
constexpr int kPieceGroups = 3;        // piece products per column: hi*hi, mid*hi, hi*mid
// MFMAs per tile pair: 3 piece products per column + the constant slots, 16 slots per MFMA
constexpr int nm_for(int n_cols) { return (kPieceGroups * n_cols + kConstSlots + 15) / 16; }
constexpr int kMaxMfma = nm_for(kMaxCols);   // 13
constexpr size_t kHdrBytes = 1024;     // word 0: max |x'|^2 (float bits, UNSCALED); word 1: non-finite flag;
                                       // words 2..3 / 4..5: evaluated-chain counters (population / neighbour sweep);
                                       // words 8..11: extent of columns 0/1; word 12: ~key of min FE, 13: key of
                                       // max finite FE; words 20..24: the scale of the current sweep (kHdrScale)
constexpr size_t kHdrSums = 256;       // byte 256..767: column sums (double) for the centring
constexpr size_t kHdrMeans = 768;      // byte 768..1023: column means as float (what x' = x - mu uses)
static_assert(kHdrSums + 8 * kMaxCols <= kHdrMeans && kHdrMeans + 4 * kMaxCols <= kHdrBytes,
              "header regions sized for kMaxCols columns");
constexpr float kNormLimit = 1.0e36f;  // larger |x'|^2 could overflow the Gram form -> flagged

// ---- components (pruned population sweeps) ------------------------------------------------------------------
// The Gram form's guard band grows with max |x - origin|^2 (DESIGN.md "guard band"), so ONE origin for a data set
// whose clusters lie far apart makes every chain wade through band pairs.  The population sweeps therefore cut the
// frames into COMPONENTS: sets that are at least rho apart in the (col 0, col 1) plane -- connected components of
// a coarse occupancy grid under "occupied boxes closer than rho".  Each component gets its own origin (centre of its
// box in columns 0/1, the column means elsewhere), its own fine cell grid, and a contiguous, group-aligned range of the
// sweep's order (pad positions carry kInvalidFrame); in the matrix-core sweep a query group only ever meets the tiles of
// its own component.  rho = r_max (sweeps that list pairs): no pair of different components can be inside any radius.
// rho = r_max / 2 (plain population sweeps): components that come closer than r_max are ADJACENT (kCompAdj), and the
// few frame pairs between adjacent components are evaluated exactly, without the matrix cores, by pop_cross_kernel
// (5M x 30: three clusters whose 2-D projections come within 0.55 of each other -- less than the largest radius, 0.65).
// One component (dense data, more than kMaxComp components, a grid too busy to label) is exactly the old
// single-origin sweep.
constexpr uint32_t kInvalidFrame = 0xFFFFFFFFu;
constexpr int kMaxComp = 64;                 // components with an origin of their own
constexpr int kCoarseDim = 128;              // coarse occupancy grid: at most kCoarseDim^2 cells
constexpr int kCoarseCells = kCoarseDim * kCoarseDim;
constexpr int kMaxOccupied = 2048;           // occupied coarse cells the labelling takes (more: one component)
constexpr int kFineSub = 4;                  // the occupancy bitmap resolves a coarse cell into kFineSub^2 sub-cells
constexpr int kMaxGroupRows = 512;           // largest query group (pop_shared_kernel: 4 waves x 4 tiles x 32 rows)
constexpr uint32_t kPadTiles = (uint32_t)kMaxComp * kMaxGroupRows / 32;   // pad positions of the order, in tiles
// layout of the component region (L.off_comp), 32-bit words
constexpr size_t kCompGrid = 0;                                        // [8]: bits(gc), -, -, ncx, ncy, n_comp
constexpr size_t kCompCellBox = 64;                                    // [cells] float4: box of the occupied sub-cells (lo0 > hi0: empty)
constexpr size_t kCompCellComp = kCompCellBox + 4 * (size_t)kCoarseCells;   // [cells]: component of the cell
constexpr size_t kCompOrigin = kCompCellComp + (size_t)kCoarseCells;   // [kMaxComp][kMaxCols] floats
constexpr size_t kCompFine = kCompOrigin + (size_t)kMaxComp * kMaxCols;      // [kMaxComp][4]: bits(min0), bits(min1), bits(cell edge 0), bits(cell edge 1)
constexpr size_t kCompNby = kCompFine + 4 * (size_t)kMaxComp;          // [kMaxComp]: cells along column 1 of the component's fine grid
constexpr size_t kCompCellOff = kCompNby + (size_t)kMaxComp;           // [kMaxComp + 1]: first cell number of a component (the cells of all components are numbered consecutively)
constexpr size_t kCompStart = kCompCellOff + (size_t)kMaxComp + 1;     // [2][kMaxComp + 1]: first sorted index of a component (reference / query order)
constexpr size_t kCompRange = kCompStart + 2 * ((size_t)kMaxComp + 1); // [2][kMaxComp + 1][2]: tile range [lo, hi) of a component (reference / query
                                                                       // order); entry kMaxComp is the empty range of the all-pad tiles at the end
constexpr size_t kCompRangeStride = 2 * ((size_t)kMaxComp + 1);
constexpr size_t kCompAdj = kCompRange + 2 * kCompRangeStride;         // [kMaxComp][2]: 64-bit mask of the components whose boxes are closer than r_max
constexpr size_t kCompBox = kCompAdj + 2 * (size_t)kMaxComp;           // [kMaxComp] float4: box of the component in columns 0/1
constexpr size_t kCompBitmap = kCompBox + 4 * (size_t)kMaxComp;        // occupancy of the sub-cells, one byte each
constexpr size_t kCompBitmapWords = (size_t)kCoarseCells * kFineSub * kFineSub / 4;
constexpr size_t kCompBase = kCompBitmap + kCompBitmapWords;           // [2][kMaxComp + 1]: first POSITION of a component in the padded order (reference / query order)
constexpr size_t kCompHash = ((kCompBase + 2 * ((size_t)kMaxComp + 1)) + 1) & ~(size_t)1;   // [64] x 64 bits: shares of the hash of the neighbour sweep's order
constexpr size_t kCompWords = kCompHash + 2 * 64;

// Workspace layout.  Regions used by the population sweep: hdr, img, norms.  The neighbour sweep
// adds a second operand image with the reference frames ORDERED BY FREE ENERGY (img_s, norms_s),
// the permutation (perm: sorted position -> frame id, invpos: frame id -> sorted position), the
// sorted free energies (fe_s, +inf padded) and, per frame, the number of frames with strictly
// lower free energy (pq) -- plus scratch for the radix sort (keys/vals double buffers; the sort's
// own temp storage sits after `fixed_end` and is sized by dc_mfma.hip).
struct Layout {
  uint32_t T, NM;
  uint32_t Tp;          // tiles of a padded order (pruned population sweeps): T + kPadTiles
  // img: A form, natural order; img_b: B form, natural order (queries of the full sweeps);
  // img_s: A form, frames ordered by free energy (full neighbour sweep)
  size_t off_img, off_img_b, off_norm, off_img_s, off_norm_s, off_fe_s, off_perm, off_invpos, off_pq,
      off_keys_in, off_keys_out, off_vals_in,
      // spatially ordered frames (2-D cell key on columns 0/1) for the pruned population sweep:
      // reference image (A form) / norms / permutation / per-tile boxes, and the same for the query
      // rows (B form)
      off_img_p, off_norm_p, off_perm_p, off_box_p, off_img_q, off_norm_q, off_perm_q, off_box_q,
      off_ferange_p,   // per reference tile (fe_lo, fe_hi) -- pruned neighbour sweep
      off_coords_p,    // ORIGINAL coordinates gathered into the reference order (exact path reads)
      off_merge64,     // [2][n_rows] packed (d2, id) for merging reference chunks (neighbour sweep)
      off_box_t,       // tile boxes regrouped by reference share (neighbour sweep: contiguous scans)
      off_comp,        // component region (kComp* words) + per-tile component of the reference / query order
      off_tile_comp, off_tile_comp_q,
      fixed_end;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline Layout make_layout(size_t n_rows, size_t n_cols) {
  Layout L;
  L.T = (uint32_t)((n_rows + 31) / 32);
  L.NM = (uint32_t)nm_for((int)n_cols);
  L.Tp = L.T + kPadTiles;
  // (every per-position region is sized for the padded orders)
  const size_t img_bytes = (size_t)16 * 64 * (size_t)L.Tp * L.NM;
  const size_t row_bytes = align256(sizeof(float) * 32 * (size_t)L.Tp);
  // (the component region sits right behind the header: ONE fill zeroes both at the start of a call)
  L.off_comp = kHdrBytes;
  L.off_img = align256(L.off_comp + sizeof(uint32_t) * kCompWords);
  L.off_img_b = align256(L.off_img + img_bytes);
  L.off_norm = align256(L.off_img_b + img_bytes);
  L.off_img_s = L.off_norm + row_bytes;
  L.off_norm_s = align256(L.off_img_s + img_bytes);
  L.off_fe_s = L.off_norm_s + row_bytes;
  L.off_perm = L.off_fe_s + row_bytes;
  L.off_invpos = L.off_perm + row_bytes;
  L.off_pq = L.off_invpos + row_bytes;
  L.off_keys_in = L.off_pq + row_bytes;
  L.off_keys_out = L.off_keys_in + row_bytes;
  L.off_vals_in = L.off_keys_out + row_bytes;
  L.off_img_p = L.off_vals_in + row_bytes;
  L.off_norm_p = align256(L.off_img_p + img_bytes);
  L.off_perm_p = L.off_norm_p + row_bytes;
  L.off_box_p = L.off_perm_p + row_bytes;
  L.off_img_q = align256(L.off_box_p + sizeof(float) * 4 * (size_t)L.Tp);
  L.off_norm_q = align256(L.off_img_q + img_bytes);
  L.off_perm_q = L.off_norm_q + row_bytes;
  L.off_box_q = L.off_perm_q + row_bytes;
  L.off_ferange_p = align256(L.off_box_q + sizeof(float) * 4 * (size_t)L.Tp);
  L.off_coords_p = align256(L.off_ferange_p + sizeof(float) * 2 * (size_t)L.Tp);
  L.off_merge64 = align256(L.off_coords_p + sizeof(float) * 32 * (size_t)L.Tp * n_cols);
  L.off_box_t = align256(L.off_merge64 + sizeof(unsigned long long) * 2 * n_rows);
  L.off_tile_comp = align256(L.off_box_t + sizeof(float) * 4 * ((size_t)L.Tp + L.Tp / 32 + 64));   // (+ one pad box per share)
  L.off_tile_comp_q = align256(L.off_tile_comp + sizeof(uint32_t) * (size_t)L.Tp);
  L.fixed_end = align256(L.off_tile_comp_q + sizeof(uint32_t) * (size_t)L.Tp);
  return L;
}

struct Ptrs {
  const uint32_t* hdr;   // [0] max norm bits, [1] non-finite flag
  const double* sums;
  const uint4* img;
  const uint4* img_b;
  const float* norms;
  const uint4* img_s;
  const float* norms_s;
  const float* fe_s;
  const uint32_t* perm;
  const uint32_t* invpos;
  const uint32_t* pq;
  const uint4* img_p;
  const float* norms_p;
  const uint32_t* perm_p;
  const float4* box_p;
  const uint4* img_q;
  const float* norms_q;
  const uint32_t* perm_q;
  const float4* box_q;
  const float* coords_p;   // original coordinates gathered into the reference order
  const uint32_t* comp;        // component region (kComp* words)
  const uint32_t* tile_comp;   // component of every tile of the reference order
  const uint32_t* tile_comp_q; // ... of the query order (row ranges)
};

inline Ptrs ws_ptrs(void* d_ws, const Layout& L) {
  char* p = (char*)d_ws;
  return Ptrs{(const uint32_t*)p,
              (const double*)(p + kHdrSums),
              (const uint4*)(p + L.off_img),
              (const uint4*)(p + L.off_img_b),
              (const float*)(p + L.off_norm),
              (const uint4*)(p + L.off_img_s),
              (const float*)(p + L.off_norm_s),
              (const float*)(p + L.off_fe_s),
              (const uint32_t*)(p + L.off_perm),
              (const uint32_t*)(p + L.off_invpos),
              (const uint32_t*)(p + L.off_pq),
              (const uint4*)(p + L.off_img_p),
              (const float*)(p + L.off_norm_p),
              (const uint32_t*)(p + L.off_perm_p),
              (const float4*)(p + L.off_box_p),
              (const uint4*)(p + L.off_img_q),
              (const float*)(p + L.off_norm_q),
              (const uint32_t*)(p + L.off_perm_q),
              (const float4*)(p + L.off_box_q),
              (const float*)(p + L.off_coords_p),
              (const uint32_t*)(p + L.off_comp),
              (const uint32_t*)(p + L.off_tile_comp),
              (const uint32_t*)(p + L.off_tile_comp_q)};
}

// ordered-integer image of a float (ascending) and back; header words 8..11 hold the bounding box of
// columns 0/1 as ~key(min0), key(max0), ~key(min1), key(max1) (all maintained with atomicMax)
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
// Cell edge of the spatial orderings: about `frames_per_cell` of the n_rows frames per cell of the
// bounding box of columns 0/1.  Small cells make the 32-frame tiles compact in that plane (a tile's
// box is about its cell), which is what the pruning lives on; a cell should still hold a few tiles.
// Measured on C3: neighbours 38.6 ms at 8192 frames per cell, 30.0 at 512, 27.4 at 128; populations
// 36.1 ms with cell = radius (about 15000 frames), 31.7 ms at about 60 frames per cell.
constexpr float kPopCellFrames = 64.0f, kNnCellFrames = 128.0f;
__device__ __forceinline__ float auto_cell(const uint32_t* __restrict__ hdr, uint32_t n_rows,
                                           float frames_per_cell) {
  const float e0 = fkey_inv(hdr[9]) - fkey_inv(~hdr[8]), e1 = fkey_inv(hdr[11]) - fkey_inv(~hdr[10]);
  const double a0 = (e0 > 0.0f && e0 <= FLT_MAX) ? (double)e0 : 0.0;
  const double a1 = (e1 > 0.0f && e1 <= FLT_MAX) ? (double)e1 : 0.0;
  const double f = (double)frames_per_cell / (double)(n_rows ? n_rows : 1u);
  const double c = (a0 > 0.0 && a1 > 0.0) ? sqrt(a0 * a1 * f) : (a0 + a1) * f;
  return (float)c;
}

// ---------------------------------------------------------------------------------------------
// scale of a SWEEP: everything the matrix pipe sees is multiplied by a factor chosen per sweep -- the centred
// coordinates by c (x'' = fl(c x')), thresholds / distances by S = fl(c c); norms are taken from the scaled
// coordinates.  Two rules (pick_scale_*), applied by scale_kernel (dc_mfma.hip), which leaves the factors in header
// words 20..24 for the image builder and the kernels:
//   neighbour sweeps   c a power of two, S M in [2^26, 2^28): exact, the pieces use the top of the fp16 range;
//   population sweeps  the largest S for which the guard band eps of the launch is <= 1 -- c is then NOT a power of
//                      two, which costs one rounding per coordinate (a few percent on the band's constants, see
//                      guard_e0_linear) and buys a band that is exactly as wide as eps: with a power of two it
//                      was up to twice that, and band pairs are most of a wide multi-radius sweep (C5: eps = 0.65
//                      of the band it paid for, 1.7 band pairs per chain).  The threshold is folded as
//                      c_q = |x''|^2 - (S r^2 - 1), so the accumulator t of a pair says
//                          t < 0        inside          (sign bit)
//                          t >= 2       outside         (bit 30: biased exponent >= 128)
//                          otherwise    band -> exact   (neither bit)
//                      and the epilogue needs ONE instruction per accumulator register: v_alignbit shifts both
//                      bits into a string (no minimum over the elements for the band test).  S M is then
//                      ~2^16 .. 2^17.6 (eps ~ 100 .. 400 u S M), the coordinates ~2^8 .. 2^9: the pieces sit in
//                      the middle of the fp16 range, and the ones that would fall below its smallest normal are
//                      kept by scaling the mid-piece products (mid 2^g x hi 2^-g, g = 6).
// M = max |x'|^2 (header word 0).
// ---------------------------------------------------------------------------------------------
struct ScaleExp {
  float c;      // coordinates (both sides) times c
  float s2;     // fl(c * c): thresholds, exact distances -> scaled units
  int g;        // mid pieces are stored as mid 2^g, their partner hi pieces as hi 2^-g
  int a;        // constant slots: 2^a on the A side, the pieces of c_q / 2^a on the B side
  int rounded;  // c is not a power of two: the scaled coordinates carry one more rounding
};
struct Scale {
  float sa, sb, s2;   // c (reference side), c (query side), fl(c c)
  float up, dn;       // 2^g, 2^-g
  float cinv;         // 2^-a
  int g, a, rounded;
};
constexpr uint32_t kHdrScale = 20;      // header words 20..24: bits(c), bits(s2), g, a, rounded
constexpr uint32_t kHdrFp = 14;         // words 14..15: content fingerprint of the array the statistics belong to (64 bits); 16..17: the
                                        // fingerprint a DC_FLAG_STATS_VALID call recomputes for the guard (dc_mfma.hip fp_term); 18..19: hash of the
                                        // neighbour sweep's order (layout header of the all-gather blocks)
constexpr uint32_t kHdrCookie = 28;     // whose statistics the header holds (array, shape); 0 after a reset
constexpr uint32_t kHdrMused = 31;      // the extent max |x - origin|^2 the current sweep's scale was chosen for (float bits)
constexpr uint32_t kHdrOpen = 30;       // neighbour sweeps: queries listed for the search in other components (nn_open_kernel)
// MFMA instructions the sweep kernels ISSUED (64-bit counters; bench.py's executed-flop figure: x 32*32*16*2 flop): header
// words 6..7 for the population sweeps, 26..27 for the neighbour sweeps -- whose early-out leaves most chains at their
// coarse MFMAs, so chains x NM would overstate it.  Addressed relative to the chain counters the kernels already get
// (words 2..3 / 4..5): + kMfmaCtrPop / + kMfmaCtrNn 64-bit words.
constexpr uint32_t kHdrMfmaPop = 6, kHdrMfmaNn = 26;
constexpr uint32_t kHdrLayoutBad = 18;     // nn_block_unpack_kernel: the gathered blocks were packed under different layouts (nothing was unpacked)
constexpr uint32_t kHdrStatsBlocks = 25;   // rows of stats_kernel's table that components_kernel has still to add up (dc_prep.hpp)
constexpr int kMfmaCtrPop = (kHdrMfmaPop - 2) / 2, kMfmaCtrNn = (kHdrMfmaNn - 4) / 2;
constexpr uint32_t kHdrShift = 32;      // population sweeps: the number of in-place threshold shifts the scale's band pays for (dc_mfma_msym.hpp; 0: none)
constexpr uint32_t kHdrMloc = 29;       // pruned population sweeps: max |x - origin(component of x)|^2 (float bits, with a rounding margin)
constexpr int kMidShiftPop = 6, kConstShiftPop = 6, kConstShiftNn = 15;
constexpr float kThrCapPop = 1048576.0f;      // 2^20 (population scale: eps <= 1 keeps S r^2 below 2^19.2 and
                                              //  4 S M below 2^19.6; only a radius beyond the clamps of
                                              //  pick_scale_pop reaches the cap, and then every pair is inside)

__host__ __device__ inline Scale make_scale(const ScaleExp& e) {
  Scale s;
  s.sa = e.c;
  s.sb = e.c;
  s.s2 = e.s2;
  s.up = ldexpf(1.0f, e.g);
  s.dn = ldexpf(1.0f, -e.g);
  s.cinv = ldexpf(1.0f, -e.a);
  s.g = e.g;
  s.a = e.a;
  s.rounded = e.rounded;
  return s;
}
__device__ __forceinline__ Scale load_scale(const uint32_t* __restrict__ hdr) {
  ScaleExp e;
  e.c = __uint_as_float(hdr[kHdrScale + 0]);
  e.s2 = __uint_as_float(hdr[kHdrScale + 1]);
  e.g = (int)hdr[kHdrScale + 2];
  e.a = (int)hdr[kHdrScale + 3];
  e.rounded = (int)hdr[kHdrScale + 4];
  return make_scale(e);
}

__host__ __device__ inline ScaleExp pick_scale_nn(float M) {
  int e = 0;
  (void)frexpf(M, &e);                  // M = f 2^e, f in [0.5, 1); M = 0 -> e = 0
  int k = (28 - e) >> 1;                // floor: e + 2k in {27, 28}
  k = k < -62 ? -62 : (k > 62 ? 62 : k);
  return ScaleExp{ldexpf(1.0f, k), ldexpf(1.0f, 2 * k), 0, kConstShiftNn, 0};
}

// ---------------------------------------------------------------------------------------------
// guard band (DESIGN.md "guard band"), in SCALED units.  For every pair, with d2 its canonical
// squared distance, u = 2^-24, M = max |x'|^2 and c_q the folded query constant (file header):
//     | acc - (d2 + c_q - |x'|^2) |  <=  e0 + kappa * (d2 + |thr|)
// where thr = r^2 for the population sweep (c_q = |x'|^2 - (r^2 - eps)) and 0 for the neighbour sweep.
//   u * [ 3 M + thr                                   norms rounded once, c_q = fl(|x'|^2 - thr)
//       + 4.1 (M + thr)                               c_q carried by two fp16 pieces
//       + 27 M                                        dropped piece products (mid*mid, hi*rho, rho*hi)
//       + 17 (2 M + thr) + (nb-1) 18 (4.02 M + thr)   the nb MFMAs that hold c_q and the hi*hi products:
//                                                     17 addends, each truncated to 2^-24 of the largest
//       + ns 18 (d2 + thr + 0.004 M) + (d2 + thr)     the ns MFMAs of small products, accumulator ~ d2 - thr
//       + (D/4 + 9) d2 + 2 M                          canonical summation order + centring (as for fp32)
//       + [rounded] 2.1 (M + thr + d2) ]              x'' = fl(c x'): |x'' - y''|^2 differs from c^2 |x' - y'|^2 by
//                                                     <= 4 u c^2 |x' - y'| sqrt(M) <= 2 u S (d2 + M); the threshold
//                                                     fl(fl(c c) r^2) from c^2 r^2 by 2 u S r^2
//   + flush                                           values below 2^-14 (the smallest normal fp16) stored as zero:
//       (2^(-12-g) + [g > 0] 2^(-23+g)) sqrt(D M)       lost piece remainders (|rho| <= max(2^-22 |v|, 2^(-14-g)))
//                                                       times the other side (sum_k |w_k| <= sqrt(D) |w|), and
//                                                       hi 2^-g copies below 2^-14 times their mid partners
//                                                       (<= 2^-11 |v|)
//       + [g > 0] D 2^(-27+g)                           the same where both factors are tiny
//       + 2^(-14+a)                                     the remainder of c_q / 2^a
// with a further factor 1.25 on everything.  nb = ceil((D + 2) / 16), ns = NM - nb.
// ---------------------------------------------------------------------------------------------
struct GuardBand {
  float e0;      // absolute part (M and thr terms)
  float kappa;   // relative part, per unit of d2
};

__host__ __device__ inline float next_up(float f) {   // f >= 0 finite; inf / NaN -> inf
  if (!(f <= FLT_MAX)) return INFINITY;
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, f) + 1u);
}

// (doubles: the population rule evaluates these at scales far from the final one)
__host__ __device__ inline double guard_e0_linear(double M, double thr, int D, int rounded, int folded = 0) {   // M, thr: scaled
  const double u = 5.9604644775390625e-8;
  const int nb = (D + kConstSlots + 15) / 16, ns = nm_for(D) - nb;
  const double t = (thr > 0.0) ? thr : 0.0;
  const double cR = rounded ? 2.1 : 0.0;
  // folded (nn_pruned_kernel: the reference norm in the constant slots, the query norm outside the accumulator): the ns
  // MFMAs of small products add to an accumulator of magnitude <= M + d2 instead of ~d2 -- 18 u M each
  const double cM = 3.0 + 4.1 + 27.0 + 34.0 + 72.4 * (nb - 1) + 0.072 * ns + 2.0 + cR + (folded ? 18.0 * ns : 0.0);
  const double cT = 1.0 + 4.1 + 17.0 + 18.0 * (nb - 1) + 18.0 * ns + 1.0 + cR;
  return 1.25 * u * (cM * M + cT * t);
}
__host__ __device__ inline double guard_flush(double M, int D, int g, int a) {
  const double fl_rel = ldexp(1.0, -12 - g) + (g > 0 ? ldexp(1.0, -23 + g) : 0.0);
  return 1.25 * (fl_rel * sqrt((double)D * M) + (g > 0 ? (double)D * ldexp(1.0, -27 + g) : 0.0) + ldexp(1.0, -14 + a));
}
__host__ __device__ inline double guard_e0(double M, double thr, int D, int g, int a, int rounded, int folded = 0) {
  return guard_e0_linear(M, thr, D, rounded, folded) + guard_flush(M, D, g, a);
}
__host__ __device__ inline double guard_kappa(int D, int rounded) {
  const double u = 5.9604644775390625e-8;
  const int nb = (D + kConstSlots + 15) / 16, ns = nm_for(D) - nb;
  return 1.25 * u * (18.0 * ns + 1.0 + 0.25 * D + 9.0 + (rounded ? 2.1 : 0.0));
}
__host__ __device__ inline GuardBand guard_band(float M, float thr, int D, const Scale& sc, bool folded = false) {   // M, thr: scaled
  GuardBand gb;
  gb.e0 = next_up((float)guard_e0((double)M, (double)thr, D, sc.g, sc.a, sc.rounded, folded ? 1 : 0));
  gb.kappa = next_up((float)guard_kappa(D, sc.rounded));
  return gb;
}

// In-place threshold shifts of the multi-radius symmetric sweep (dc_mfma_msym.hpp): radius k's threshold is taken off the
// accumulator by ONE MFMA, ones x (three fp16 pieces of -delta_k), steps of them in a row.  A step adds C and three exact
// products: by the hardware fact of guard_e0 its four addends are truncated to q <= u max|addend| and the sum is rounded
// once -- |error| <= (4 + 1/2) u max(|acc|, |delta|).  For a pair whose value at its own radius is within the band's
// reach (|t| <= 2 + the error itself) every earlier accumulator value and every delta is <= span + 4 (span: the largest
// difference between two scaled thresholds of the launch, here bounded by the largest one); the steps themselves are
// float differences of float deltas (two roundings, <= u span each step: the thresholds the steps add up to are that far
// from the ones the exact path compares with); pieces below 2^-14 are stored as zero (subnormal MFMA inputs are not
// exact): 2^-14 per piece.  With the factor 1.25 of the other terms.
__host__ __device__ inline double guard_shift(double span, int steps) {   // span: scaled
  const double u = 5.9604644775390625e-8;
  return (steps <= 0) ? 0.0 : 1.25 * steps * (5.5 * u * (span + 4.0) + 3.0 * ldexp(1.0, -14));
}
// population sweep: one band for all pairs with d2 up to the largest radius of the launch (scaled units)
__host__ __device__ inline double guard_eps_pop(double M, double r2max, int D, int g, int a, int rounded, int shift_steps = 0) {
  const double cap = (r2max > 0.0) ? r2max : 0.0;
  return (guard_e0(M, r2max, D, g, a, rounded) + guard_kappa(D, rounded) * cap + guard_shift(cap, shift_steps)) * (1.0 + 1.2e-7);
}

// the population rule: the largest S with guard_eps_pop(S M, S r2max) <= 1 (M, r2max unscaled; r2max the largest
// squared radius of the call, so that the images serve every radius of it).  Every term of the band grows at most
// linearly with S, so S = 2^K / eps(2^K) is admissible when 2^K is; c = sqrt(S) rounded down, s2 = fl(c c).
__host__ __device__ inline ScaleExp pick_scale_pop(float M_in, float r2_in, int D, int shift_steps = 0) {
  const double M = (M_in > 0.0f) ? (double)M_in : 0.0;
  const double r2 = (r2_in > 0.0f) ? (double)r2_in : 0.0;   // (+inf allowed)
  constexpr int kLo = -120, kHi = 120;
  auto eps_at = [&](double S) { return guard_eps_pop(S * M, S * r2, D, kMidShiftPop, kConstShiftPop, 1, shift_steps); };
  int K = kHi;
  // eps(S) >= S * lin: an upper bound for K, lowered until the flush part fits as well (a step or two)
  const double lin = (guard_e0_linear(M, r2, D, 1) + guard_kappa(D, 1) * r2 + (shift_steps > 0 ? 1.25 * shift_steps * 5.5 * 5.9604644775390625e-8 * r2 : 0.0)) * (1.0 + 1.2e-7);
  if (!(lin <= 1.7e308)) {
    K = kLo;
  } else if (lin > 0.0) {
    int e = 0;
    (void)frexp(lin, &e);               // lin in [2^(e-1), 2^e): S lin <= 1 needs K <= 1 - e
    K = 1 - e;
    K = K < kLo ? kLo : (K > kHi ? kHi : K);
  }
  while (K > kLo && !(eps_at(ldexp(1.0, K)) <= 1.0)) --K;
  ScaleExp e;
  e.g = kMidShiftPop;
  e.a = kConstShiftPop;
  const double epsK = eps_at(ldexp(1.0, K));
  if (K <= kLo || K >= kHi || !(epsK > 0.0) || !(epsK <= 1.0)) {
    // the clamps (degenerate data, a radius beyond everything): an even power of two, exact scaling
    K &= ~1;
    e.c = ldexpf(1.0f, K / 2);
    e.s2 = ldexpf(1.0f, K);
    e.rounded = 0;
    return e;
  }
  float c = (float)sqrt(ldexp(1.0, K) / epsK * (1.0 - 1e-6));
  while (!(eps_at((double)c * (double)c) <= 1.0)) c = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, c) - 1u);
  e.c = c;
  e.s2 = (float)((double)c * (double)c);
  e.rounded = 1;
  return e;
}

// ---- fp32 (scaled) -> two fp16 pieces: v = hi + mid + rho, |rho| <= max(2^-22 |v|, 2^(-14-g)) ----------
// The mid piece is stored as mid 2^g (exact, and normal down to 2^(-14-g)), the hi piece a second time as
// hi 2^-g for the products with the other side's mid piece (zero below 2^-14: part of the band).
constexpr float kF16MinNormal = 6.103515625e-05f;   // 2^-14
__host__ __device__ inline uint32_t f16_rne(float f) {   // |f| <= 65504; below 2^-14 -> 0
  if (!(fabsf(f) >= kF16MinNormal)) return 0u;
  return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);
}
__host__ __device__ inline float f16_val(uint32_t b) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)b);
}
struct Pieces {
  uint32_t hi, mid;   // fp16 bit patterns: hi, mid 2^g
  uint32_t hi_dn;     // hi 2^-g
};
__host__ __device__ inline Pieces split2(float v, float up = 1.0f, float dn = 1.0f) {
  Pieces p;
  p.hi = f16_rne(v);
  const float hv = f16_val(p.hi);
  const float r1 = v - hv;                 // exact
  p.mid = f16_rne(r1 * up);
  p.hi_dn = f16_rne(hv * dn);              // (11 significant bits: exact unless it falls below 2^-14)
  return p;
}
__host__ __device__ inline uint32_t const_a_bits(int a) { return (uint32_t)(a + 15) << 10; }   // fp16 2^a, -14 <= a <= 15

// K-slot s of a frame: which piece of which column (or the constant) sits there.
//   slots 0..1            constant: A side 2^a, B side the pieces (hi, mid) of c_q / 2^a
//   slots 2 + G*D + k     column k, piece pair G (A piece x B piece), large products first:
//                         0 hi x hi, 1 mid 2^g x hi 2^-g, 2 hi 2^-g x mid 2^g
//   beyond 2 + 3 D        zero padding
// Returns the fp16 pattern for the A form (reference side) or the B form (query side, column values
// are those of -2x'') of a row whose centred, SCALED columns are fetched through `col(k)`.
template <class ColFn>
__host__ __device__ inline uint32_t slot_value(uint32_t s, uint32_t D, bool b_form, const Scale& sc, ColFn col) {
  if (s < (uint32_t)kConstSlots) return b_form ? 0u : const_a_bits(sc.a);   // (the kernels patch c_q in)
  const uint32_t sp = s - kConstSlots, G = sp / D, k = sp - G * D;
  if (G >= (uint32_t)kPieceGroups) return 0u;
  const float v = b_form ? -2.0f * col(k) : col(k);
  const Pieces p = split2(v, sc.up, sc.dn);
  if (G == 0u) return p.hi;
  // mid piece on the A side in group 1, on the B side in group 2; the other side holds hi 2^-g
  const bool mid = b_form ? (G == 2u) : (G == 1u);
  return mid ? p.mid : p.hi_dn;
}

template <int NM>
__device__ __forceinline__ void load_tile(const uint4* __restrict__ img,
                                          const float* __restrict__ norms, uint32_t t, int lane,
                                          int h, s16x8 (&a)[NM], float4 (&nv)[4]) {
  const uint4* ip = img + (size_t)t * (NM * 64) + lane;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const uint4 v = ip[m * 64];
    a[m] = __builtin_bit_cast(s16x8, v);
  }
  const float4* np = reinterpret_cast<const float4*>(norms + (size_t)t * 32 + 4 * h);
#pragma unroll
  for (int g = 0; g < 4; ++g) nv[g] = np[2 * g];        // rows 8g + 4h .. +3  <->  registers 4g .. 4g+3
}

// single-buffered operands (many MFMAs per chain: no room for two buffers): fragment m of the NEXT
// reference tile replaces fragment m of the current one right after the last MFMA that reads it
// (the last chain of the tile); the row norms go with fragment 0
template <int NM, int MI>
__device__ __forceinline__ void refill_frag(const uint4* __restrict__ img,
                                            const float* __restrict__ norms, uint32_t t_next,
                                            int lane, int h, s16x8 (&a)[NM], float4 (&nv)[4]) {
  const uint4 v = img[(size_t)t_next * (NM * 64) + MI * 64 + lane];
  a[MI] = __builtin_bit_cast(s16x8, v);
  if constexpr (MI == 0) {
    const float4* np = reinterpret_cast<const float4*>(norms + (size_t)t_next * 32 + 4 * h);
#pragma unroll
    for (int g = 0; g < 4; ++g) nv[g] = np[2 * g];
  }
}

// resident query-side operand of one query tile: B form fragments with the pieces of the per-lane
// constant c_q (scaled units, |c_q| <= 65504 * 2^15) patched into slots 0..1 (held by the h = 0 half)
template <int NM>
__device__ __forceinline__ void load_query(const uint4* __restrict__ img_b, uint32_t tile, int lane,
                                           int h, float cq, const Scale& sc, s16x8 (&b)[NM]) {
  const uint4* ip = img_b + (size_t)tile * (NM * 64) + lane;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const uint4 v = ip[m * 64];
    b[m] = __builtin_bit_cast(s16x8, v);
  }
  const Pieces p = split2(cq * sc.cinv);
  if (h == 0) {
    b[0][0] = (short)p.hi;
    b[0][1] = (short)p.mid;
  }
}

__device__ __forceinline__ void load_frag(const float* __restrict__ rowvals, uint32_t t, int h,
                                          float4 (&v)[4]) {
  const float4* p = reinterpret_cast<const float4*>(rowvals + (size_t)t * 32 + 4 * h);
#pragma unroll
  for (int g = 0; g < 4; ++g) v[g] = p[2 * g];
}

__device__ __forceinline__ f32x16 frag16(const float4 (&v)[4]) {
  f32x16 o;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    o[4 * g + 0] = v[g].x;
    o[4 * g + 1] = v[g].y;
    o[4 * g + 2] = v[g].z;
    o[4 * g + 3] = v[g].w;
  }
  return o;
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma16(const s16x8& a, const s16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b),
                                                c, 0, 0, 0);
}

// The row norms of a reference tile are the C operand of the FIRST MFMA of every chain on that tile.  hipcc
// picks the two-address form (vdst = srcC) for the one chain at which that operand dies -- the last of the
// tile -- and, the destination being the loop-carried accumulator, pays for it with a 16-register copy per
// tile (8 v_mov_b64).  An empty asm that reads the operand after the last chain keeps it alive, so every
// chain gets the three-address form.
__device__ __forceinline__ void keep_alive(const f32x16& v) { asm volatile("" ::"v"(v)); }

template <int NM>
__device__ __forceinline__ f32x16 gram_chain(const s16x8 (&a)[NM], const s16x8 (&b)[NM],
                                             const f32x16& c0) {
  f32x16 acc = mfma16(a[0], b[0], c0);
#pragma unroll
  for (int m = 1; m < NM; ++m) acc = mfma16(a[m], b[m], acc);
  return acc;
}

// f(integral_constant<0>), f(integral_constant<2>), ... for the even indices below N (compile-time
// indices for register arrays inside the pipelined tile bodies)
template <int N, int I = 0, class F>
__device__ __forceinline__ void constexpr_for_all(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    constexpr_for_all<N, I + 1>(f);
  }
}
template <int N, int I = 0, class F>
__device__ __forceinline__ void constexpr_for_pairs(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    constexpr_for_pairs<N, I + 2>(f);
  }
}

// reference operands of the pruned sweeps: two register buffers while they are small (<= 4 MFMAs per
// chain), one buffer refilled during the last chain of a tile beyond that
template <int NM>
constexpr bool kSingleBuffer = NM > 4;

// a query lane that owns no live row: the constant (the largest the two slots can carry, 65504 * 2^a:
// ~2^31 at the neighbour scale, ~2^22 at the population scale -- above 4 S M + 2 in both) keeps its
// accumulators above every threshold
__device__ __forceinline__ float dead_const(const Scale& sc) { return ldexpf(65504.0f, sc.a); }

// row of reference tile t held by register r of a lane in half h
__device__ __forceinline__ uint32_t tile_row(uint32_t t, int r, int h) {
  return 32u * t + (uint32_t)((r & 3) + 8 * (r >> 2) + 4 * h);
}

// canonical d2 of two rows of the ORIGINAL coordinate matrix (the only arithmetic that decides)
__device__ __attribute__((noinline)) float exact_d2(const float* __restrict__ coords,
                                                   uint32_t n_cols, uint32_t jq, uint32_t i) {
  return dist2_canon_rt(coords + (size_t)jq * n_cols, 1, coords + (size_t)i * n_cols, 1,
                        (int)n_cols);
}

// minimum of elements [R0, R1) of an accumulator tile (v_min3_f32: two elements per instruction)
template <int R0, int R1>
__device__ __forceinline__ void tile_min(const f32x16& acc, float& m) {
  if constexpr (R1 - R0 >= 2) {
    m = fminf(fminf(m, acc[R0]), acc[R0 + 1]);
    tile_min<R0 + 2, R1>(acc, m);
  } else if constexpr (R1 - R0 == 1) {
    m = fminf(m, acc[R0]);
  }
}

// =============================================================================================
// population count
// =============================================================================================
// The accumulator of radius 0 is t_0 = acc (threshold folded into c_q: c_q = |x'|^2 - (r_0^2 - 1) at the population
// scale); radius r uses t_r = acc - delta_r with delta_r = r_r^2 - r_0^2, the same for every query.  Of every t the
// epilogue keeps two bits: the sign (inside) and bit 30 (t >= 2: outside); neither = band.
template <int NR>
struct PopQ {            // per query tile, per lane
  uint32_t cnt[NR];
};

constexpr uint32_t kSignBits = 0xAAAAAAAAu;   // the sign of element r sits at bit 31 - 2 r of a chain's string
constexpr uint32_t kBandKey = 0x40000000u;    // bits(2.0f): band <=> bits(t) <u kBandKey

template <int NR>
struct PopAcc {          // per chain scratch
  uint32_t bits[NR];     // (sign, bit 30) of t_r, two bits per element, shifted in from the right
};

template <int NR>
struct PopDeltas {
  float d[NR];           // d[0] is 0 and never used
};

template <int NR>
__device__ __forceinline__ PopDeltas<NR> pop_deltas(const Rad2& rad2e) {
  PopDeltas<NR> o;
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) o.d[rr] = rad2e.v[rr] - rad2e.v[0];
  return o;
}

template <int NR>
__device__ __forceinline__ void pop_epi_begin(PopAcc<NR>& e) {
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) e.bits[rr] = 0;
}

// elements [R0, R1) of one accumulator tile (element r ends up at bits 31 - 2 r, 30 - 2 r of the strings)
template <int NR, int R0, int R1>
__device__ __forceinline__ void pop_epi(const f32x16& acc, const PopDeltas<NR>& dl, PopAcc<NR>& e) {
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) {
#pragma unroll
    for (int r = R0; r < R1; ++r) {
      const uint32_t tb = __float_as_uint(rr == 0 ? acc[r] : acc[r] - dl.d[rr]);
      e.bits[rr] = __builtin_amdgcn_alignbit(e.bits[rr], tb, 30);   // (bits << 2) | top two bits of t
    }
  }
}
// of a chain's string: elements inside (bit 31 - 2 r) / in the band (the same positions)
__device__ __forceinline__ uint32_t inside_of(uint32_t bits) { return bits & kSignBits; }
__device__ __forceinline__ uint32_t band_of(uint32_t bits) { return ~(bits | (bits << 1)) & kSignBits; }
__device__ __forceinline__ int element_of(int bit) { return (31 - bit) >> 1; }   // bit 31 - 2 r -> r

// MFMA chain into acc_new with the epilogue of acc_old spread between the MFMAs: a wave issues in
// order, so the VALU work has to sit in the shadow of the matrix pipe in PROGRAM order
struct NoRefill {
  template <int MI>
  __device__ __forceinline__ void operator()(std::integral_constant<int, MI>) const {}
};

template <int NM, int NR, int MI = 0, class Refill = NoRefill>
__device__ __forceinline__ void pop_chain(const s16x8 (&a)[NM], const s16x8 (&b)[NM],
                                          const f32x16& c0, f32x16& acc_new,
                                          const f32x16& acc_old, const PopDeltas<NR>& dl,
                                          PopAcc<NR>& e, const Refill& refill = Refill{}) {
  if constexpr (MI < NM) {
    if constexpr (MI == 0)
      acc_new = mfma16(a[0], b[0], c0);
    else
      acc_new = mfma16(a[MI], b[MI], acc_new);
    refill(std::integral_constant<int, MI>{});   // fragment MI of the next tile (last chain only)
    pop_epi<NR, (16 * MI) / NM, (16 * (MI + 1)) / NM>(acc_old, dl, e);
    pop_chain<NM, NR, MI + 1, Refill>(a, b, c0, acc_new, acc_old, dl, e, refill);
  }
}

template <int NR>
struct PopDelta {
  uint32_t d[NR];
  unsigned long long key;   // kSinkMinEdge: lightest outgoing pair found among the evaluated entries
};

// rare: exact re-check of the band pairs of one accumulator tile.  Everything by value and a
// returned delta, so that neither the accumulators nor the per-query state ever get an address
// (an escaping reference would park them in scratch for the whole hot loop).
template <int NR>
__device__ __attribute__((noinline)) PopDelta<NR> pop_fix(const float* __restrict__ coords,
                                                          const uint32_t* __restrict__ perm,
                                                          uint32_t n_rows, uint32_t n_cols,
                                                          Rad2 rad2, PopDeltas<NR> dl, f32x16 acc,
                                                          uint32_t wbits, uint32_t jq, uint32_t t,
                                                          int h) {
  PopDelta<NR> out;
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) out.d[rr] = 0;
  uint32_t m = 0;   // elements of this lane that sit in some radius' band
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    bool any = false;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)   // the same arithmetic as pop_epi
      any = any | (__float_as_uint(rr == 0 ? acc[r] : acc[r] - dl.d[rr]) < wbits);
    m |= (any & (tile_row(t, r, h) < n_rows)) ? (1u << r) : 0u;
  }
  // lane-parallel exact evaluation: one band pair per lane per iteration (row fetches overlap)
  while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
    if (m != 0) {
      const int r = __builtin_ctz(m);
      const uint32_t pos = tile_row(t, r, h);
      const uint32_t i = perm ? perm[pos] : pos;   // reference rows may be spatially re-ordered
      const float d2c = exact_d2(coords, n_cols, jq, i);
      float av = acc[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) av = (r == k) ? acc[k] : av;   // acc[r], r per lane
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
        if (__float_as_uint(rr == 0 ? av : av - dl.d[rr]) < wbits)
          out.d[rr] += (d2c < rad2.v[rr]) ? 1u : 0u;
      m &= m - 1;
    }
  }
  return out;
}

// thresholds of a population launch, shared by the full and the pruned sweep
template <int NR>
struct PopSetup {
  Scale sc;
  uint32_t wbits;        // band <=> bits(t) <u wbits: bits(2.0f) (the band is [0, 2) in scaled units)
  Rad2 rad2e;            // r^2 - 1 (scaled)
  PopDeltas<NR> dl;
};

template <int NR>
__device__ __forceinline__ PopSetup<NR> pop_setup(const uint32_t* __restrict__ hdr, const Rad2& rad2,
                                                  uint32_t n_cols) {
  (void)n_cols;
  PopSetup<NR> P;
  // everything here is in the scaled units of the operand images (exact powers of two); the scale was chosen
  // (pick_scale_pop, scale_kernel) so that the guard band of the call's largest radius is at most 1
  P.sc = load_scale(hdr);
  P.wbits = kBandKey;
#pragma unroll
  for (int rr = 0; rr < kMaxRadiiPerLaunch; ++rr)
    P.rad2e.v[rr] = fminf(rad2.v[rr] * P.sc.s2, kThrCapPop) - 1.0f;
  // (delta_r = fl(rad2e_r - rad2e_0) adds at most u * r2max to the band: inside the 1.25 factor)
  P.dl = pop_deltas<NR>(P.rad2e);
  return P;
}

template <int NM, int NR, int TQ>
__global__ __launch_bounds__(256, 2) void pop_mfma_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const uint4* __restrict__ img, const uint4* __restrict__ img_b,
    const float* __restrict__ norms, const uint32_t* __restrict__ hdr, uint32_t T, uint32_t i_from,
    uint32_t i_to, Rad2 rad2, int n_rad, uint32_t* __restrict__ pops) {
  if (hdr[1] != 0) return;   // non-finite / overflow-prone data: the gated direct kernel runs instead
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;   // whole wave leaves; no barriers in this kernel

  const PopSetup<NR> P = pop_setup<NR>(hdr, rad2, n_cols);

  s16x8 b[TQ][NM];
  PopQ<NR> q[TQ];
  uint32_t jq[TQ];
  uint64_t livemask[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    jq[qt] = tile * 32 + c;
    const bool live = (tile < T) && (jq[qt] >= i_from) && (jq[qt] < i_to);
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    const uint32_t tl = tile < T ? tile : T - 1;
    const float cq = live ? norms[tl * 32 + c] - P.rad2e.v[0] : dead_const(P.sc);
    load_query<NM>(img_b, tl, lane, h, cq, P.sc, b[qt]);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) q[qt].cnt[rr] = 0;
  }

  s16x8 a0[NM], a1[NM];
  float4 n0[4], n1[4];
  load_tile<NM>(img, norms, 0, lane, h, a0, n0);

  // the rest of an epilogue: counts, band test, rare exact path
  auto finish = [&](const f32x16& acc, auto qi_c, const PopAcc<NR>& e, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) q[qi].cnt[rr] += __builtin_popcount(inside_of(e.bits[rr]));
    uint32_t band = 0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) band |= band_of(e.bits[rr]);
    if (__builtin_expect((__builtin_amdgcn_ballot_w64(band != 0) & livemask[qi]) != 0, 0)) {
      const PopDelta<NR> dl =
          pop_fix<NR>(coords, nullptr, n_rows, n_cols, rad2, P.dl, acc, P.wbits, jq[qi], t, h);
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) q[qi].cnt[rr] += ((livemask[qi] >> lane) & 1) ? dl.d[rr] : 0u;
    }
  };

  // chains software-pipelined over two accumulator tiles (see pop_pruned_kernel); accB holds the chain
  // whose epilogue is pending (+inf everywhere = contributes nothing)
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
  uint32_t tB = 0;
  auto tile_body = [&](const s16x8 (&a)[NM], const float4 (&nv)[4], uint32_t t) {
    const f32x16 c0 = frag16(nv);
    if constexpr (TQ == 1) {
      PopAcc<NR> e;
      pop_epi_begin<NR>(e);
      pop_chain<NM, NR>(a, b[0], c0, accA, accB, P.dl, e);
      finish(accB, std::integral_constant<int, 0>{}, e, tB);
      accB = accA;   // (16 moves against >= 9 MFMAs per chain)
    } else {
      constexpr_for_pairs<TQ>([&](auto qt_c) {
        constexpr int qt = decltype(qt_c)::value;
        constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
        PopAcc<NR> e;
        pop_epi_begin<NR>(e);
        pop_chain<NM, NR>(a, b[qt], c0, accA, accB, P.dl, e);
        finish(accB, std::integral_constant<int, qb>{}, e, (qt == 0) ? tB : t);
        pop_epi_begin<NR>(e);
        pop_chain<NM, NR>(a, b[qt + 1], c0, accB, accA, P.dl, e);
        finish(accA, std::integral_constant<int, qt>{}, e, t);
      });
    }
    keep_alive(c0);
    tB = t;
  };

  for (uint32_t t = 0; t < T; t += 2) {
    load_tile<NM>(img, norms, (t + 1 < T) ? t + 1 : t, lane, h, a1, n1);
    tile_body(a0, n0, t);
    if (t + 1 < T) {
      load_tile<NM>(img, norms, (t + 2 < T) ? t + 2 : t + 1, lane, h, a0, n0);
      tile_body(a1, n1, t + 1);
    }
  }
  {  // drain: epilogue of the last pending chain
    PopAcc<NR> e;
    pop_epi_begin<NR>(e);
    pop_epi<NR, 0, 16>(accB, P.dl, e);
    finish(accB, std::integral_constant<int, TQ - 1>{}, e, tB);
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const bool live = (livemask[qt] >> lane) & 1;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      const uint32_t total = q[qt].cnt[rr] + (uint32_t)__shfl_xor((int)q[qt].cnt[rr], 32, 64);
      if (h == 0 && live && rr < n_rad) {
        // the sweep met the self pair and counted it iff d2(i,i) < rad2; the reference starts at 1
        const float dself = exact_d2(coords, n_cols, jq[qt], jq[qt]);
        pops[(size_t)rr * n_rows + jq[qt]] = total + 1u - ((dself < rad2.v[rr]) ? 1u : 0u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// population count over SPATIALLY ORDERED frames with tile-pair pruning -- the GPU counterpart of the
// reference's box grid (density_clustering.cpp:41-89,160-168: 2-D cells on columns 0/1, only the 3x3
// neighbour cells are searched).  Frames are sorted by their cell key, so a tile of 32 consecutive
// frames is compact in the (col 0, col 1) plane and carries a bounding box; a (query tile, reference
// tile) pair whose boxes are at least r_max apart cannot hold a pair with d2 < r^2 (d2 in D
// dimensions >= d2 in two of them) and is skipped without any MFMA.  Each wave scans all reference
// boxes (64 per step) against the box of its query group, compacts the survivors into an LDS list
// and runs the usual Gram-chain + epilogue on them; per chain one more box test against the single
// query tile.  Counting, guard band and exact re-check are those of pop_mfma_kernel.
// ---------------------------------------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs by their linear id (x fastest), each XCD with its own L2.  Query
// groups are numbered along the spatial ordering, and neighbouring groups walk nearly the same reference tiles: every
// XCD gets a CONTIGUOUS eighth of the groups, so that the tiles its waves stream are shared through its L2 instead of
// being fetched by all eight.  The eighths are ROTATED from one reference share (blockIdx.y) to the next: the work of
// a group follows the local density and, in the symmetric population sweep, its place in its cluster (the groups at
// the start of a cluster carry most of its pairs), so an XCD that kept the same eighth for every share would finish
// long before or after the others.  (Round 2 had this by accident -- its grids were not multiples of 8, which shifted
// the XCD of a block from share to share; a grid of 5272 = 8 * 659 groups then ran the C3 sweep in 18 instead of 12.4
// ms with a third of the chip idle in the tail.)  The launchers round gridDim.x up to a multiple of 8, so block (x, y)
// runs on XCD x & 7; n_blocks: the real number of block-sized units.  Returns the unit of this block, 0xFFFFFFFF for a
// pad block.
__device__ __forceinline__ uint32_t xcd_block(uint32_t n_blocks) {
  const uint32_t b = blockIdx.x, phys = b & 7u, idx = b >> 3;
  const uint32_t eighth = (phys + blockIdx.y) & 7u;
  const uint32_t n_full = n_blocks >> 3, rem = n_blocks & 7u;
  if (idx >= n_full + (eighth < rem ? 1u : 0u)) return 0xFFFFFFFFu;
  return eighth * n_full + min(eighth, rem) + idx;
}
inline uint32_t grid_x8(uint32_t n_blocks) { return (n_blocks + 7u) & ~7u; }

// which rows a pruned sweep answers for, and where their operands live:
//   kQueryOwnOrder  a row range [i_from, i_to): its own spatial ordering (img_q / norms_q / perm_q / box_q)
//   kQueryAll       rows in the reference order (the reference arrays double as query arrays): all of
//                   them, or the query groups of one segment of a sharded run (QSeg)
enum QueryMode { kQueryOwnOrder = 0, kQueryAll = 1 };

// reference tiles scanned per round (LDS list entries per wave).  512, not more: with the query rows and the
// queues a block then stays below 80 KB of LDS -- two blocks per CU -- for every column count up to 64
// (1024: D = 25, 26 and 57..64 fell to one block per CU; 300k x 26: 7.4 / 8.1 ms -> 5.8 / 6.7 ms; C3 unchanged)
constexpr int kListCap = 512;
constexpr int kQueueCap = 4;     // deferred exact evaluations: entries per lane and query tile
constexpr int kWaveQueue = 128;  // ... or per WAVE, in one compact list (flushed in batches of 64)
constexpr int kSeedNeighbours = 4;   // neighbour sweep: frames on either side of a query evaluated up front

// ---- deferred exact re-check (pruned sweep) --------------------------------------------------------
// A band pair is not evaluated the moment it appears (one or two lanes busy for a full, dependent
// memory latency each time -- measured: a third of the sweep); it is parked as (reference position,
// radii whose band it sits in) in a small per-lane LDS queue.  When a lane's queue is full, and at
// the end, all lanes evaluate their entries in parallel: query row from LDS (staged once per wave),
// reference row from a copy of the ORIGINAL coordinates gathered into the reference order.
constexpr uint32_t kPopQueuePosBits = 24;    // entry = position | radius flags << 24
constexpr uint32_t kPopQueueMaxRows = 1u << kPopQueuePosBits;
static_assert(kMinEdgeMaxRows + kOrderPadRows == kPopQueueMaxRows, "min-edge sweeps take the queue path only");
static_assert(kOrderPadRows == (size_t)32 * kPadTiles, "pad positions of the orders");

// wave-aggregated append: lanes with `have` add one pair each
__device__ __forceinline__ void emit_edge(const EdgeSink& sink, bool have, uint32_t pos_q, uint32_t pos_r) {
  const uint64_t m = __builtin_amdgcn_ballot_w64(have);
  if (m == 0) return;
  unsigned long long base = 0;
  const int leader = __builtin_ctzll(m);
  if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(sink.count, (unsigned long long)__builtin_popcountll(m));
  base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader);
  const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
  if (have && sink.edges && base + rank < sink.capacity) sink.edges[base + rank] = make_uint2(pos_q, pos_r);
}

// candidate of the lightest outgoing pair of a query's component (kSinkMinEdge)
__device__ __forceinline__ unsigned long long min_edge_key(const EdgeSink& sink, uint32_t comp_q,
                                                           uint32_t rank_q, uint32_t pos_r) {
  const uint32_t comp_r = sink.comp[pos_r];
  if (comp_r == comp_q) return ~0ull;
  const uint32_t rank_r = sink.rank[pos_r];
  return ((unsigned long long)max(rank_q, rank_r) << 32) | min(rank_q, rank_r);
}

template <int NR, int MODE>
__device__ __attribute__((noinline)) PopDelta<NR> pop_flush(const uint32_t* queue /* [kQueueCap][64] */,
                                                            uint32_t count, const float* qrow,
                                                            const float* __restrict__ coords_r,
                                                            uint32_t n_cols, Rad2 rad2, int lane,
                                                            EdgeSink sink, uint32_t pos_q,
                                                            uint32_t comp_q, uint32_t rank_q) {
  PopDelta<NR> out;
  out.key = ~0ull;
#pragma unroll
  for (int rr = 0; rr < NR; ++rr) out.d[rr] = 0;
#pragma unroll
  for (int k = 0; k < kQueueCap; ++k) {
    if (__builtin_amdgcn_ballot_w64((uint32_t)k < count) == 0) break;
    if ((uint32_t)k < count) {
      const uint32_t ent = queue[k * 64 + lane];
      const uint32_t pos = ent & (kPopQueueMaxRows - 1u), flags = ent >> kPopQueuePosBits;
      const float d2c = dist2_canon_rows(qrow, coords_r + (size_t)pos * n_cols, (int)n_cols);
#pragma unroll
      for (int rr = 0; rr < NR; ++rr)
        out.d[rr] += ((((flags >> rr) & 1u) != 0u) & (d2c < rad2.v[rr])) ? 1u : 0u;
      if constexpr (MODE == kSinkPairs) emit_edge(sink, (d2c < rad2.v[0]) & (pos < pos_q), pos_q, pos);
      if constexpr (MODE == kSinkMinEdge)
        if (d2c < rad2.v[0]) out.key = min(out.key, min_edge_key(sink, comp_q, rank_q, pos));
    }
  }
  return out;
}

// Wave-wide deferred exact re-check of the plain population sweep (one radius, no pair sink).  The per-lane
// queues above flush when ONE lane is full, with most lanes idle, and every flush walks kQueueCap slots; here
// the wave appends (reference position, query) to ONE compact list and evaluates 64 entries at a time, one
// per lane.  The owner of the query is credited through an LDS counter.  Measured at C3 (A/B): -0.7 % (the
// flushes are a small part there; at C5's radii, in pop_shared_kernel, the same change took 222 -> 204 ms).
// (A build without any band handling runs the C3 sweep in 18.1 instead of 22.5 ms -- but mostly because the
// compiler then also drops the 8 v_min3_u32 of the band DETECTION from every chain, not because of this path.)
// sym_tq > 0 (symmetric sweep): a pair whose reference tile lies outside the wave's own group of sym_tq query tiles
// is evaluated by this wave alone, so the reference frame is credited as well (pops_pos: counts by position).
__device__ __attribute__((noinline)) void pop_wave_flush_rows(const uint32_t* queue, uint32_t qn,
                                                              const float* qrows, uint32_t* fix_tab,
                                                              const float* __restrict__ coords_r,
                                                              uint32_t n_cols, float r2, int lane,
                                                              uint32_t* __restrict__ pops_pos = nullptr,
                                                              uint32_t sym_tq = 0, uint32_t own_group = 0) {
  for (uint32_t k0 = 0; k0 < qn; k0 += 64) {
    if (k0 + lane < qn) {
      const uint32_t ent = queue[k0 + lane];
      const uint32_t pos = ent & (kPopQueueMaxRows - 1u), qidx = ent >> kPopQueuePosBits;
      const float d2c = dist2_canon_rows(qrows + (size_t)qidx * n_cols, coords_r + (size_t)pos * n_cols, (int)n_cols);
      if (d2c < r2) {
        atomicAdd(&fix_tab[qidx], 1u);
        if (sym_tq != 0u && (pos >> 5) / sym_tq != own_group) atomicAdd(&pops_pos[pos], 1u);
      }
    }
  }
}

// ---- symmetric population sweep: the reference side of a chain -------------------------------------------------
// d2(i, j) = d2(j, i) (the canonical sum squares differences: the same value in either order), so a tile pair
// needs one chain, not two, if BOTH frames of an inside pair are credited -- the reference's own trick
// (density_clustering.cpp:170, 179-182).  The query side of a chain is an in-lane popcount; the reference side is
// a sum ACROSS the 32 query lanes of a half-wave, per reference row.  It is taken once per reference tile, over
// the TQ chains of the wave's query tiles together (they meet the same 32 reference rows):
//   in-lane     the sign bits of three strings add up bit-sliced (xor3 / majority: two v_bitop3) into 2-bit fields,
//               the fields of the groups of three are added in 4-bit slots (even / odd elements apart);
//   lanes       one step of the lane reduction fits the 4-bit slots (<= 12), then the slots are spread to bytes
//               (four words, <= 192 = TQ * 32) for the remaining steps: row_shr 2, 4, 8 and row_bcast15 into the
//               odd rows -- lanes 31 / 63 end up with the sums over the queries of their half;
//   rows        those two lanes leave their 16 bytes in LDS, lane i < 32 picks the byte of row i and adds it to
//               the row's count with one 128-byte atomic per tile.
// ~50 VALU instructions per reference tile, i.e. 8 per chain at TQ = 6 -- against the 21 + 2 MFMAs of the second
// chain they replace.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_take(uint32_t v) {   // v of the source lane, 0 where there is none
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
}
// byte (of the 32 staged by lanes 31 / 63) that holds the count of reference row i = lane, i < 32
__device__ __forceinline__ uint32_t ref_credit_byte(int lane) {
  const int i = lane & 31, hh = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3), f = 15 - r;
  return (uint32_t)(16 * hh + 4 * (f & 3) + (f >> 2));
}
template <int TQ>
__device__ __forceinline__ void ref_credit(const uint32_t (&sb)[TQ], uint32_t t, uint32_t n_rows,
                                           uint32_t* __restrict__ pops_pos, uint32_t* stage /* 8 words of LDS */,
                                           uint32_t my_byte, int lane) {
  static_assert(TQ <= 6, "4-bit slots hold the sums of two groups of three strings over two lanes");
  {  // nothing inside in the whole tile (tiles at the edge of the pruning radius): nothing to credit
    uint32_t any = 0;
#pragma unroll
    for (int q = 0; q < TQ; ++q) any |= sb[q];
    if (__builtin_amdgcn_ballot_w64((any & kSignBits) != 0u) == 0) return;
  }
  uint32_t A = 0, B = 0;
#pragma unroll
  for (int g = 0; g < TQ; g += 3) {
    const uint32_t a = sb[g], b = (g + 1 < TQ) ? sb[g + 1] : 0u, c = (g + 2 < TQ) ? sb[g + 2] : 0u;
    const uint32_t lo = a ^ b ^ c, hi = (a & b) | (a & c) | (b & c);   // valid at the sign positions (odd bits)
    const uint32_t x = ((lo >> 1) & 0x55555555u) | (hi & kSignBits);   // element r: 0..3 at bits 31-2r, 30-2r
    A += x & 0x33333333u;
    B += (x >> 2) & 0x33333333u;
  }
  A += dpp_take<0x111>(A);   // row_shr:1
  B += dpp_take<0x111>(B);
  uint32_t W[4] = {A & 0x0F0F0F0Fu, B & 0x0F0F0F0Fu, (A >> 4) & 0x0F0F0F0Fu, (B >> 4) & 0x0F0F0F0Fu};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    W[e] += dpp_take<0x112>(W[e]);         // row_shr:2
    W[e] += dpp_take<0x114>(W[e]);         // row_shr:4
    W[e] += dpp_take<0x118>(W[e]);         // row_shr:8
    W[e] += dpp_take<0x142, 0xA>(W[e]);    // row_bcast15 into rows 1 and 3: lanes 31 / 63 hold their half's sums
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) u32x4 LdsU4;
  typedef __attribute__((address_space(3))) unsigned char LdsU8;
  if ((lane & 31) == 31) ((LdsU4*)stage)[lane >> 5] = u32x4{W[0], W[1], W[2], W[3]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (one wave: its LDS operations execute in order)
  __builtin_amdgcn_wave_barrier();
  const uint32_t cnt = ((const volatile LdsU8*)stage)[my_byte];
  __builtin_amdgcn_wave_barrier();
  const uint32_t row = 32u * t + (uint32_t)lane;
  if (lane < 32 && cnt != 0u && row < n_rows) atomicAdd(&pops_pos[row], cnt);
}

__device__ __forceinline__ float box_gap2(const float4& a, const float4& b) {
  // boxes are (lo0, hi0, lo1, hi1); squared distance between them in the (col 0, col 1) plane
  const float dx = fmaxf(0.0f, fmaxf(a.x - b.y, b.x - a.y));
  const float dy = fmaxf(0.0f, fmaxf(a.z - b.w, b.z - a.w));
  return dx * dx + dy * dy;
}

// what a pruned population sweep needs to know about the components (file header of this section): the component of
// every QUERY tile, the tile range of every component in the REFERENCE order, and the number of positions of that order
struct CompView {
  const uint32_t* tile_comp_q;
  const uint32_t* range_r;   // [kMaxComp][2]
  uint32_t n_pos;
  const uint32_t* comp;      // the component region itself (fine grids, boxes)
};

// SYM: the symmetric sweep (all rows as queries, one radius, no sink): a query group meets its own tiles as
// before and, of the other groups, the half that lies ahead of it on the circle of groups -- every unordered pair
// of groups once -- crediting both sides (ref_credit); all counts go to pops_pos (by position, zero-filled).
// Query rows of one tile (original coordinates, for the exact path) into LDS [32][n_cols]: from the ordered copy of the
// rows -- contiguous, coalesced -- when the queries are rows of the reference order, gathered by frame otherwise; four
// loads per lane in flight either way (a load per trip of a run-time column loop waited for each: ten round trips at the
// start of every wave; a quarter less memory-side traffic, kernel times within the noise of a box).
__device__ __forceinline__ void stage_query_rows(float* __restrict__ dst, const float* __restrict__ contig,
                                                 const float* __restrict__ coords, uint32_t frame, bool live,
                                                 uint32_t n_cols, int lane) {
  if (contig) {
    const uint32_t total = 32u * n_cols;
    for (uint32_t e0 = (uint32_t)lane; e0 < total; e0 += 256u) {
      float v[4];
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) v[j] = (e0 + 64u * j < total) ? contig[e0 + 64u * j] : 0.0f;
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j)
        if (e0 + 64u * j < total) dst[e0 + 64u * j] = v[j];
    }
  } else if (lane < 32) {
    const float* x = coords + (size_t)frame * n_cols;
    for (uint32_t k0 = 0; k0 < n_cols; k0 += 4) {
      float v[4];
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) v[j] = (live && k0 + j < n_cols) ? x[k0 + j] : 0.0f;
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j)
        if (k0 + j < n_cols) dst[(uint32_t)lane * n_cols + k0 + j] = v[j];
    }
  }
}

template <int NM, int NR, int TQ, int MODE = kSinkNone, bool SYM = false>
__global__ __launch_bounds__(256, 2) void pop_pruned_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const uint4* __restrict__ img_r, const float* __restrict__ norms_r,
    const uint32_t* __restrict__ perm_r, const float4* __restrict__ box_r,
    const float* __restrict__ coords_r, uint32_t T,
    const uint4* __restrict__ img_q, const float* __restrict__ norms_q,
    const uint32_t* __restrict__ perm_q, const float4* __restrict__ box_q, uint32_t n_q, QSeg q_seg,
    const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ chain_counter, Rad2 rad2,
    int n_rad, uint32_t* __restrict__ pops, EdgeSink sink, CompView CV, uint32_t* __restrict__ pops_pos = nullptr) {
  static_assert(!SYM || (MODE == kSinkNone && NR == 1), "the symmetric sweep is the plain one-radius sweep");
  __shared__ uint32_t credit_stage[4][8];
  // dynamic LDS, per wave of the workgroup (one wave, see nn_pruned_kernel): the survivor list of a scan round
  // [kListCap], then [TQ*32][n_cols] query rows (original coordinates), then the queues of deferred exact
  // evaluations [TQ][kQueueCap][64]
  extern __shared__ __attribute__((aligned(16))) float pop_dyn_lds[];
  if (hdr[1] != 0) return;   // flagged data: the gated direct kernel runs instead
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const int wib = threadIdx.x >> 6;
  const uint32_t wpb = blockDim.x >> 6;
  uint32_t* lists_all = reinterpret_cast<uint32_t*>(pop_dyn_lds);
  float* pop_qrows_all = pop_dyn_lds + (size_t)wpb * kListCap;
  const uint32_t TQT = (n_q + 31) / 32;
  const uint32_t blk_unit = xcd_block((seg_groups((TQT + TQ - 1) / TQ, q_seg) + wpb - 1) / wpb);
  if (blk_unit == 0xFFFFFFFFu) return;   // (pad block of the grid)
  const uint32_t wave = seg_group(blk_unit * wpb + wib, q_seg);
  // gridDim.y > 1: the reference tiles are dealt round-robin to gridDim.y waves per query group and
  // the partial counts are merged with atomics (keeps small launches, e.g. one rank of an 8-GPU
  // run, at >= 2 waves per SIMD without giving up the operand reuse of TQ query tiles per wave)
  const uint32_t chunk = blockIdx.y, n_chunks = gridDim.y;
  const uint32_t qt0 = wave * TQ;
  if (qt0 >= TQT) return;    // whole wave leaves; no block-level barriers in this kernel
  // (the component of the group and its tile range: two dependent look-ups, started before everything else)
  const uint32_t my_comp = CV.tile_comp_q[qt0];
  const uint32_t comp_lo = CV.range_r[2 * my_comp], comp_hi = CV.range_r[2 * my_comp + 1];
  uint32_t* list = lists_all + (size_t)wib * kListCap;
  float* qrows = pop_qrows_all + (size_t)wib * (TQ * 32) * n_cols;
  uint32_t* queues = reinterpret_cast<uint32_t*>(pop_qrows_all + (size_t)wpb * (TQ * 32) * n_cols) +
                     (size_t)wib * TQ * kQueueCap * 64;
  const bool use_queue = CV.n_pos <= kPopQueueMaxRows;   // positions fit the queue entries
  // plain sweep of one radius: ONE compact list per wave (carved out of the same LDS region: 128 entries and a
  // counter per query instead of TQ x kQueueCap x 64 entries)
  constexpr bool kWaveWide = (MODE == kSinkNone) && (NR == 1) && (TQ * 32 <= 256) && (TQ * kQueueCap * 64 >= kWaveQueue + TQ * 32);
  uint32_t* fix_tab = queues + kWaveQueue;   // [TQ*32]: band pairs of a query that the exact path found inside
  uint32_t qn = 0;                           // queued entries (wave-uniform)
  uint32_t qcount[TQ];
  // kSinkMinEdge: component and rank of this lane's query, lightest outgoing pair seen so far
  uint32_t comp_q[TQ], rank_q[TQ];
  unsigned long long edge_key[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    qcount[qt] = 0;
    comp_q[qt] = 0;
    rank_q[qt] = 0;
    edge_key[qt] = ~0ull;
  }

  const PopSetup<NR> P = pop_setup<NR>(hdr, rad2, n_cols);
  float r2max = rad2.v[0];
#pragma unroll
  for (int rr = 1; rr < NR; ++rr) r2max = fmaxf(r2max, rad2.v[rr]);
  const float far2 = r2max * 1.0001f;   // boxes at least this far apart (squared) hold no pair inside

  s16x8 b[TQ][NM];
  PopQ<NR> q[TQ];
  uint32_t jq[TQ];
  uint64_t livemask[TQ];
  float4 qbox[TQ];
  float4 gbox = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    const uint32_t tl = tile < TQT ? tile : TQT - 1;
    const uint32_t pos = tile * 32 + c;
    // (pad positions of the order -- the components are padded to whole query groups -- carry kInvalidFrame)
    const uint32_t frame = ((tile < TQT) && (pos < n_q)) ? perm_q[pos] : kInvalidFrame;
    const bool live = frame != kInvalidFrame;
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    jq[qt] = live ? frame : 0u;
    const float cq = live ? norms_q[tl * 32 + c] - P.rad2e.v[0] : dead_const(P.sc);
    load_query<NM>(img_q, tl, lane, h, cq, P.sc, b[qt]);
    if constexpr (MODE == kSinkMinEdge) {   // (all rows, in the reference order: position = pos)
      comp_q[qt] = live ? sink.comp[pos] : 0xFFFFFFFFu;
      rank_q[qt] = live ? sink.rank[pos] : 0u;
    }

#pragma unroll
    for (int rr = 0; rr < NR; ++rr) q[qt].cnt[rr] = 0;
    qbox[qt] = (tile < TQT) ? box_q[tile] : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    gbox.x = fminf(gbox.x, qbox[qt].x);
    gbox.y = fmaxf(gbox.y, qbox[qt].y);
    gbox.z = fminf(gbox.z, qbox[qt].z);
    gbox.w = fmaxf(gbox.w, qbox[qt].w);
  }
  // the original coordinates of the queries (for the exact path) into LDS -- after the loop, so that the loads of all the
  // tiles above are issued together (nn_pruned_kernel: a store per tile in between cost a fifth of a wave's set-up)
  if (perm_q == perm_r && qt0 + TQ <= TQT) {
    // (the rows of the wave's tiles are one contiguous piece of the ordered copy)
    const float* src = coords_r + (size_t)qt0 * 32 * n_cols;
    const uint32_t total = (uint32_t)TQ * 32u * n_cols;
    for (uint32_t e0 = (uint32_t)lane; e0 < total; e0 += 1280u) {
      float v[20];
#pragma unroll
      for (uint32_t j = 0; j < 20; ++j) v[j] = (e0 + 64u * j < total) ? src[e0 + 64u * j] : 0.0f;
#pragma unroll
      for (uint32_t j = 0; j < 20; ++j)
        if (e0 + 64u * j < total) qrows[e0 + 64u * j] = v[j];
    }
  } else {
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      const uint32_t tile = qt0 + qt, tl = tile < TQT ? tile : TQT - 1;
      stage_query_rows(qrows + (size_t)qt * 32 * n_cols, (perm_q == perm_r) ? coords_r + (size_t)tl * 32 * n_cols : nullptr,
                       coords, jq[qt], (livemask[qt] >> lane) & 1, n_cols, lane);
    }
  }
  if constexpr (kWaveWide) {
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt)
      if (h == 0) fix_tab[qt * 32 + c] = 0;
  }

  // SYM: groups of TQ tiles on a circle; this wave's group and its count; the strings of the chains on the pending
  // reference tile and whether that tile is credited (it lies outside the wave's own group)
  const uint32_t n_groups = (TQT + TQ - 1) / TQ;
  uint32_t sb[TQ];
  bool symB = false;
  const uint32_t my_byte = ref_credit_byte(lane);
  auto flush_wave = [&]() {
    if constexpr (SYM)
      pop_wave_flush_rows(queues, qn, qrows, fix_tab, coords_r, n_cols, rad2.v[0], lane, pops_pos, (uint32_t)TQ, wave);
    else
      pop_wave_flush_rows(queues, qn, qrows, fix_tab, coords_r, n_cols, rad2.v[0], lane);
    qn = 0;
  };
  // evaluate and empty the queue of query tile qi (all lanes in parallel per slot)
  auto flush = [&](int qi) {
    if constexpr (kWaveWide) return;
    if (__builtin_amdgcn_ballot_w64(qcount[qi] != 0) == 0) return;
    const PopDelta<NR> dl =
        pop_flush<NR, MODE>(queues + qi * (kQueueCap * 64), qcount[qi], qrows + (qi * 32 + c) * n_cols,
                            coords_r, n_cols, rad2, lane, sink, (qt0 + (uint32_t)qi) * 32u + (uint32_t)c,
                            comp_q[qi], rank_q[qi]);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) q[qi].cnt[rr] += dl.d[rr];
    if constexpr (MODE == kSinkMinEdge) edge_key[qi] = min(edge_key[qi], dl.key);
    qcount[qi] = 0;
  };

  uint32_t chains = 0;
  // this wave's share of the reference tiles: t = chunk + u * n_chunks, u = 0 .. U-1 (round-robin, so
  // every share sees every region; the scan only touches its own boxes)
  // ... of the tiles of the group's own COMPONENT: every other frame is at least r_max away (CompView)
  const uint32_t t_lo = comp_lo, t_hi = min(comp_hi, T);
  const uint32_t u_lo = (t_lo > chunk) ? (t_lo - chunk + n_chunks - 1) / n_chunks : 0u;
  const uint32_t U = (t_hi > chunk) ? (t_hi - chunk + n_chunks - 1) / n_chunks : 0u;
  for (uint32_t base = u_lo; base < U; base += kListCap) {
    // ---- scan: which reference tiles of this round can hold a pair within r_max of the group?
    uint32_t cnt = 0;
    const uint32_t lim = min(U - base, (uint32_t)kListCap);
    auto tile_of = [&](uint32_t u) { return chunk + u * n_chunks; };
    // (the box of step k+1 is fetched while step k is tested: the scan is latency-, not work-bound)
    float4 rb_next = ((uint32_t)lane < lim) ? box_r[tile_of(base + lane)]
                                            : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    for (uint32_t k = 0; k < lim; k += 64) {
      const uint32_t t = tile_of(base + k + lane);
      const float4 rb = rb_next;
      if (k + 64 + lane < lim) rb_next = box_r[tile_of(base + k + 64 + lane)];
      // (one test against the box of the whole query group: per-query-tile masks were measured to
      //  save < 0.5 % of the chains and they keep the chains from being pipelined)
      bool ok = (k + lane < lim) && (box_gap2(gbox, rb) < far2);
      if constexpr (SYM) {
        // the wave's own group, or a group at most half the circle ahead (exactly half: the lower index takes it)
        const uint32_t gt = t / (uint32_t)TQ;
        const uint32_t ahead = (gt >= wave) ? gt - wave : gt + n_groups - wave;
        ok = ok & ((2u * ahead < n_groups) | ((2u * ahead == n_groups) & (wave < gt)));
      }
      const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
      if (ok) list[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = t;
      cnt += (uint32_t)__builtin_popcountll(m);
    }
    if (cnt == 0) continue;
    // ---- process the survivors.  The operands of survivor i+1 are in flight while survivor i is
    //      computed, and the chains are software-pipelined over two accumulator tiles: while the
    //      MFMAs of one chain run, the epilogue of the previous chain issues in their shadow.
    s16x8 a0[NM];
    float4 n0[4];
    auto entry = [&](uint32_t i) {
      return (uint32_t)__builtin_amdgcn_readfirstlane(list[i < cnt ? i : cnt - 1]);
    };
    // the rest of an epilogue: counts, band test, parking of the band pairs
    auto finish = [&](const f32x16& acc, auto qi_c, const PopAcc<NR>& e, uint32_t t) {
      constexpr int qi = decltype(qi_c)::value;
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) q[qi].cnt[rr] += __builtin_popcount(inside_of(e.bits[rr]));
      if constexpr (SYM) {   // (the band of a whole reference tile is looked at once, from the kept strings: park_tile)
        sb[qi] = e.bits[0];
        return;
      }
      if constexpr (MODE == kSinkMinEdge) {
        // partners decided "inside" by the accumulator alone (element r = bit 31 - 2 r of the
        // string; the query itself is one of them and belongs to its own component): keep the
        // lightest pair that leaves the component.  Inside elements are sparse (< 1 % at 4 sigma^2).
        uint32_t inside = inside_of(e.bits[0]);
        while (inside != 0) {
          const int p = __builtin_ctz(inside);
          edge_key[qi] = min(edge_key[qi], min_edge_key(sink, comp_q[qi], rank_q[qi], tile_row(t, element_of(p), h)));
          inside &= inside - 1u;
        }
      }
      if constexpr (MODE == kSinkPairs) {
        // pairs decided "inside" by the accumulator alone: element r = bit 31 - 2 r of the string.
        // Only partners at a smaller position are listed (every pair once); one atomic per chain and
        // wave reserves the slots, then each lane stores its own pairs.
        const uint32_t pos_q = (qt0 + (uint32_t)qi) * 32u + (uint32_t)c;
        uint32_t inside = inside_of(e.bits[0]);
        if (32u * t + 31u >= pos_q) {       // (all rows of earlier tiles are smaller: nothing to mask)
          uint32_t keep = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) keep |= (tile_row(t, r, h) < pos_q) ? (0x80000000u >> (2 * r)) : 0u;
          inside &= keep;
        }
        const uint32_t k = (uint32_t)__builtin_popcount(inside);
        if (__builtin_amdgcn_ballot_w64(k != 0) != 0) {
          uint32_t incl = k;                // inclusive prefix sum over the wave
#pragma unroll
          for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
            incl += (lane >= off) ? v : 0u;
          }
          unsigned long long base = 0;
          if (lane == 63) base = atomicAdd(sink.count, (unsigned long long)incl);
          base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(base >> 32), 63) << 32) |
                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, 63);
          unsigned long long idx = base + (incl - k);
          while (inside != 0) {
            const int p = __builtin_ctz(inside);
            if (idx < sink.capacity) sink.edges[idx] = make_uint2(pos_q, tile_row(t, element_of(p), h));
            ++idx;
            inside &= inside - 1u;
          }
        }
      }
      uint32_t fl[NR];   // per radius: bit (31 - 2 r) set <=> element r sits in that radius' band
      uint32_t m = 0;
#pragma unroll
      for (int rr = 0; rr < NR; ++rr) {
        fl[rr] = band_of(e.bits[rr]);
        m |= fl[rr];
      }
      if (__builtin_expect((__builtin_amdgcn_ballot_w64(m != 0) & livemask[qi]) != 0, 0)) {
        if (use_queue) {
          // park the band elements of this lane: (position, radii whose band holds the element).  Pad rows
          // (acc = +inf) and idle lanes (acc ~ 2^22) are never in a band, so no further masking is needed.
          if constexpr (kWaveWide) {
            for (;;) {
              const uint64_t have = __builtin_amdgcn_ballot_w64(m != 0);
              if (have == 0) break;
              const uint32_t n_new = (uint32_t)__builtin_popcountll(have);
              if (qn + n_new > (uint32_t)kWaveQueue) flush_wave();
              if (m != 0) {
                const int p = __builtin_ctz(m);
                const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
                queues[slot] = tile_row(t, element_of(p), h) | ((uint32_t)(qi * 32 + c) << kPopQueuePosBits);
                m &= m - 1;
              }
              qn += n_new;
            }
            if (qn >= 64u) flush_wave();
            return;
          }
          uint32_t* qu = queues + qi * (kQueueCap * 64);
          while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
            if (__builtin_amdgcn_ballot_w64((m != 0) & (qcount[qi] == (uint32_t)kQueueCap)) != 0)
              flush(qi);
            if (m != 0) {
              const int p = __builtin_ctz(m);
              uint32_t flags = 0;
#pragma unroll
              for (int rr = 0; rr < NR; ++rr) flags |= ((fl[rr] >> p) & 1u) << rr;
              qu[qcount[qi] * 64 + lane] = tile_row(t, element_of(p), h) | (flags << kPopQueuePosBits);
              ++qcount[qi];
              m &= m - 1;
            }
          }
        } else {
          const PopDelta<NR> dl =
              pop_fix<NR>(coords, perm_r, CV.n_pos, n_cols, rad2, P.dl, acc, P.wbits, jq[qi], t, h);
#pragma unroll
          for (int rr = 0; rr < NR; ++rr) q[qi].cnt[rr] += ((livemask[qi] >> lane) & 1) ? dl.d[rr] : 0u;
        }
      }
    };
    // SYM: the band pairs of reference tile t, from the strings of its TQ chains -- one test per tile instead of one
    // per chain (a compare and a scalar branch less in every chain)
    auto park_tile = [&](uint32_t t) {
      uint32_t decided = 0xFFFFFFFFu;
#pragma unroll
      for (int qi = 0; qi < TQ; ++qi) decided &= sb[qi] | (sb[qi] << 1);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64((~decided & kSignBits) != 0u) == 0, 1)) return;
#pragma unroll
      for (int qi = 0; qi < TQ; ++qi) {
        uint32_t m = band_of(sb[qi]);
        for (;;) {
          const uint64_t have = __builtin_amdgcn_ballot_w64(m != 0);
          if (have == 0) break;
          const uint32_t n_new = (uint32_t)__builtin_popcountll(have);
          if (qn + n_new > (uint32_t)kWaveQueue) flush_wave();
          if (m != 0) {
            const int p = __builtin_ctz(m);
            const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
            queues[slot] = tile_row(t, element_of(p), h) | ((uint32_t)(qi * 32 + c) << kPopQueuePosBits);
            m &= m - 1;
          }
          qn += n_new;
        }
      }
      if (qn >= 64u) flush_wave();
    };
    bool pendB = false;   // SYM: sb[] holds the strings of a real reference tile (tB)
    // accB always holds the chain whose epilogue is still pending: query tile TQ-1 of reference
    // tile tB (or +inf everywhere = contributes nothing)
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
    uint32_t tB = 0;
    // t_next: the survivor after t (single-buffered operands are refilled during the last chain)
    auto compute = [&](s16x8 (&a)[NM], float4 (&nv)[4], uint32_t t, uint32_t t_next) {
      const f32x16 c0 = frag16(nv);
      chains += TQ;
      static_assert(TQ % 2 == 0, "accumulator ping-pong needs an even number of query tiles");
      auto refill = [&](auto mi_c) {
        if constexpr (kSingleBuffer<NM>)
          refill_frag<NM, decltype(mi_c)::value>(img_r, norms_r, t_next, lane, h, a, nv);
      };
      constexpr_for_pairs<TQ>([&](auto qt_c) {
        constexpr int qt = decltype(qt_c)::value;
        constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
        PopAcc<NR> e;
        pop_epi_begin<NR>(e);
        pop_chain<NM, NR>(a, b[qt], c0, accA, accB, P.dl, e);
        finish(accB, std::integral_constant<int, qb>{}, e, (qt == 0) ? tB : t);
        if constexpr (SYM && qt == 0) {   // the strings of tile tB are complete now
          if (pendB) park_tile(tB);
          if (symB) ref_credit<TQ>(sb, tB, CV.n_pos, pops_pos, credit_stage[wib], my_byte, lane);
        }
        pop_epi_begin<NR>(e);
        if constexpr (qt + 2 == TQ)   // last chain of the tile
          pop_chain<NM, NR>(a, b[qt + 1], c0, accB, accA, P.dl, e, refill);
        else
          pop_chain<NM, NR>(a, b[qt + 1], c0, accB, accA, P.dl, e);
        finish(accA, std::integral_constant<int, qt>{}, e, t);
      });
      keep_alive(c0);
      tB = t;
      symB = (t / (uint32_t)TQ) != wave;
      pendB = true;
    };
    if constexpr (kSingleBuffer<NM>) {
      uint32_t e0 = entry(0);
      load_tile<NM>(img_r, norms_r, e0, lane, h, a0, n0);
      for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t e1 = entry(i + 1);
        compute(a0, n0, e0, e1);
        e0 = e1;
      }
    } else {
      s16x8 a1[NM];
      float4 n1[4];
      // (the survivor list is read one tile ahead of its use, as in nn_pruned_kernel: the LDS latency sat in front of
      //  every tile's loads)
      auto peek = [&](uint32_t i) { return list[i < cnt ? i : cnt - 1]; };
      uint32_t e0 = entry(0), e1;
      uint32_t l_next = peek(1);
      load_tile<NM>(img_r, norms_r, e0, lane, h, a0, n0);
      for (uint32_t i = 0; i < cnt; i += 2) {
        e1 = (uint32_t)__builtin_amdgcn_readfirstlane(l_next);
        l_next = peek(i + 2);
        load_tile<NM>(img_r, norms_r, e1, lane, h, a1, n1);
        compute(a0, n0, e0, e1);
        if (i + 1 < cnt) {
          e0 = (uint32_t)__builtin_amdgcn_readfirstlane(l_next);
          l_next = peek(i + 3);
          load_tile<NM>(img_r, norms_r, e0, lane, h, a0, n0);
          compute(a1, n1, e1, e0);
        }
      }
    }
    {  // drain: epilogue of the last pending chain of this round
      PopAcc<NR> e;
      pop_epi_begin<NR>(e);
      pop_epi<NR, 0, 16>(accB, P.dl, e);
      finish(accB, std::integral_constant<int, TQ - 1>{}, e, tB);
      if constexpr (SYM) {
        if (pendB) park_tile(tB);
        if (symB) ref_credit<TQ>(sb, tB, CV.n_pos, pops_pos, credit_stage[wib], my_byte, lane);
        symB = false;
        pendB = false;
      }
    }
  }
  if (lane == 0 && chain_counter) {
    atomicAdd(chain_counter, (unsigned long long)chains);
    atomicAdd(chain_counter + kMfmaCtrPop, (unsigned long long)chains * NM);
  }
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) flush(qt);
  if constexpr (kWaveWide) flush_wave();
  if constexpr (MODE == kSinkMinEdge) {
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt)
      if (edge_key[qt] != ~0ull) atomicMin(&sink.best[comp_q[qt]], edge_key[qt]);
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const bool live = (livemask[qt] >> lane) & 1;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      uint32_t total = q[qt].cnt[rr] + (uint32_t)__shfl_xor((int)q[qt].cnt[rr], 32, 64);
      if constexpr (kWaveWide) total += fix_tab[qt * 32 + c];
      if (h == 0 && live && rr < n_rad) {
        // the sweep met the self pair (box gap 0: never pruned) and counted it iff d2(i,i) < rad2;
        // the reference starts every population at 1 (:132-134): corrected once, by chunk 0
        uint32_t v = total;
        if (chunk == 0) {
          const float dself = exact_d2(coords, n_cols, jq[qt], jq[qt]);
          v += 1u - ((dself < rad2.v[rr]) ? 1u : 0u);
        }
        if constexpr (SYM)
          atomicAdd(&pops_pos[(qt0 + (uint32_t)qt) * 32u + (uint32_t)c], v);   // (all rows: position = place in the order)
        else if (n_chunks == 1)
          pops[(size_t)rr * n_rows + jq[qt]] = v;
        else
          atomicAdd(&pops[(size_t)rr * n_rows + jq[qt]], v);   // pops was zero-filled by the caller
      }
    }
  }
}


#include "dc_mfma_shared.hpp"
#include "dc_mfma_msym.hpp"

// =============================================================================================
// nearest neighbour / nearest neighbour with lower free energy
// =============================================================================================
__device__ __forceinline__ void lexi_update(bool cond, float& bd, uint32_t& bj, float d,
                                            uint32_t j, uint32_t n_rows) {
  // strict '<' on d2 scanning ascending j  ==  lexicographic min on (d2, j); a tie only counts
  // against a REAL incumbent (the initial FLT_MAX / n_rows+1 never loses a tie: :257-260, :270).
  // Written as selects on purpose: the branchy form of this update was mis-structurised by
  // hipcc 7.2 (the tie winner's index move was dropped), caught by the duplicate-frame test.
  const bool take = cond & ((d < bd) | ((d == bd) & (j < bj) & (bj <= n_rows)));
  bd = take ? d : bd;
  bj = take ? j : bj;
}

// The reference frames of this sweep are ORDERED BY FREE ENERGY (ascending).  For a query with
// pq frames of strictly lower free energy, "fe[j] < fe[i]" (:275) is simply "sorted position < pq":
// whole tiles are below (hd minimum = nn minimum of the tile), above (contribute nothing) or -- one
// tile per query -- straddle the boundary.  The straddling tile and the tile that holds the query
// itself (i != j, :262) take a per-element masked epilogue; every other tile costs 8 v_min3 and a
// handful of scalar-ish ops per 1024 pairs.
struct NnQ {             // per query tile, per lane
  float m_nn, m_hd;      // running minima of the MFMA values (this lane's reference rows only)
  float bd_nn, bd_hd;    // exact incumbents
  uint32_t bj_nn, bj_hd;
  uint32_t pq;           // number of frames with strictly lower free energy
  uint32_t spos;         // sorted position of the query itself
  uint32_t t_self;       // reference tile that holds the query itself
  uint32_t t_full;       // reference tiles [0, t_full) lie entirely below the query's free energy
  uint32_t t_part;       // the tile straddling the boundary (0xFFFFFFFF if pq is a multiple of 32)
};

struct NnBest {
  float bd_nn, bd_hd;
  uint32_t bj_nn, bj_hd;
};

struct NnMin {
  float tmin, hmin;
};

// general (per-element) minima of one accumulator tile: self excluded, hd restricted to pos < pq
__device__ __attribute__((noinline)) NnMin nn_special(f32x16 acc, uint32_t t, int h, uint32_t spos,
                                                      uint32_t pq) {
  NnMin o{INFINITY, INFINITY};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t pos = tile_row(t, r, h);
    const float v = (pos != spos) ? acc[r] : INFINITY;
    o.tmin = fminf(o.tmin, v);
    o.hmin = fminf(o.hmin, (pos < pq) ? v : INFINITY);
  }
  return o;
}

// rare: exact evaluation of the candidates of one accumulator tile (values within the band of the
// running minimum), merged lexicographically on (d2, frame id)
__device__ __attribute__((noinline)) NnBest nn_fix(const float* __restrict__ coords,
                                                   const uint32_t* __restrict__ perm,
                                                   uint32_t n_rows, uint32_t n_cols, f32x16 acc,
                                                   float bn, float bh, NnBest best, uint32_t jq,
                                                   uint32_t spos, uint32_t pq, uint32_t t, int h) {
  uint32_t mn = 0, mh = 0;   // candidate elements of this lane
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t pos = tile_row(t, r, h);
    const bool other = (pos != spos) & (pos < n_rows);
    mn |= (other & (acc[r] < bn)) ? (1u << r) : 0u;
    mh |= (other & (pos < pq) & (acc[r] < bh)) ? (1u << r) : 0u;
  }
  uint32_t m = mn | mh;      // lane-parallel exact evaluation, one candidate per lane per iteration
  while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
    if (m != 0) {
      const int r = __builtin_ctz(m);
      const uint32_t j = perm[tile_row(t, r, h)];
      const float d2c = exact_d2(coords, n_cols, jq, j);
      lexi_update((mn >> r) & 1u, best.bd_nn, best.bj_nn, d2c, j, n_rows);
      lexi_update((mh >> r) & 1u, best.bd_hd, best.bj_hd, d2c, j, n_rows);
      m &= m - 1;
    }
  }
  return best;
}

// band of the neighbour sweep: candidates are the frames whose MFMA value is below
// m + 2.5 (e0 + kappa m), m the running minimum (the relative part makes the band follow the
// distance scale of the query; 2.5 > 2 covers the error of the minimum AND of the candidate)
__device__ __forceinline__ float nn_band(const GuardBand& g, float m) {
  return m + 2.5f * (g.e0 + g.kappa * fmaxf(m, 0.0f));
}

// MFMA chain into acc_new with the tile minimum of acc_old spread between the MFMAs (see pop_chain)
template <int NM, int MI = 0, class Refill = NoRefill>
__device__ __forceinline__ void nn_chain(const s16x8 (&a)[NM], const s16x8 (&b)[NM],
                                         const f32x16& c0, f32x16& acc_new, const f32x16& acc_old,
                                         float& tmin, const Refill& refill = Refill{}) {
  if constexpr (MI < NM) {
    if constexpr (MI == 0)
      acc_new = mfma16(a[0], b[0], c0);
    else
      acc_new = mfma16(a[MI], b[MI], acc_new);
    refill(std::integral_constant<int, MI>{});   // fragment MI of the next tile (last chain only)
    tile_min<(16 * MI) / NM, (16 * (MI + 1)) / NM>(acc_old, tmin);
    nn_chain<NM, MI + 1, Refill>(a, b, c0, acc_new, acc_old, tmin, refill);
  }
}

// ---- early-out of the pruned neighbour sweep -----------------------------------------------------------------
// The K slots are ordered "large products first" (slot_value): the constant and every hi x hi product sit in the
// first kNnCoarse<NM> MFMAs of a chain, the mid x hi / hi x mid corrections after them.  Each correction is at most
// 2^-11 (1 + 2^-11) |a_k| * |hi(2 b_k)|, all of them together at most 2^-9 (1 + 2^-10) |a| |b| <= 2^-9 (1 + 2^-10) M
// (M: the extent max |x'|^2 of the sweep, scaled), and the accumulation of the remaining MFMAs moves the value by
// less than 2^-16 M on top (the guard band's own bound for it is 18 * 2^-24 M per MFMA).  So after the coarse
// part every FINAL element is >= coarse - nn_skip_bound(M): when the coarse tile minimum minus that bound is not
// below the chain's candidate threshold in any lane, the rest of the chain can change neither a running minimum
// nor a candidate list, and is skipped.  Chains that are not skipped run the very MFMA sequence they always ran (the
// per-wave kernel starts them again from their first MFMA, the shared-operand kernel goes on from the coarse part).
template <int NM>
constexpr int kNnCoarse = (NM <= 2) ? 1 : 2;          // n_cols <= 10: 12 slots; n_cols <= 20: 22 slots (NM <= 4)
template <int NM>
constexpr bool kNnEarly = !kSingleBuffer<NM> && NM > kNnCoarse<NM>;   // (two operand buffers, and something to skip)
__device__ __forceinline__ float nn_skip_bound(float M_scaled) {
  return M_scaled * (0.001953125f * 1.01f + 1.52587890625e-05f);   // 2^-9 * 1.01 + 2^-16
}

// the coarse part of a chain (NB MFMAs; `a` holds at least those fragments) into acc_new, with the tile minimum of
// acc_old (a coarse accumulator too) in its shadow
template <int NM, int NB, int NA, int MI = 0>
__device__ __forceinline__ void nn_chain_coarse(const s16x8 (&a)[NA], const s16x8 (&b)[NM], const f32x16& c0,
                                                f32x16& acc_new, const f32x16& acc_old, float& tmin) {
  static_assert(NB <= NA && NB <= NM, "coarse fragments");
  if constexpr (MI < NB) {
    if constexpr (MI == 0)
      acc_new = mfma16(a[0], b[0], c0);
    else
      acc_new = mfma16(a[MI], b[MI], acc_new);
    tile_min<(16 * MI) / NB, (16 * (MI + 1)) / NB>(acc_old, tmin);
    nn_chain_coarse<NM, NB, NA, MI + 1>(a, b, c0, acc_new, acc_old, tmin);
  }
}
// the MFMAs of the coarse part that hold the constant and every hi x hi product: ceil((n_cols + 2) / 16), and the most
// an NM can need (n_cols <= (16 NM - 2) / 3)
__host__ __device__ constexpr int nn_coarse_for(int n_cols) { return (n_cols + kConstSlots + 15) / 16; }
template <int NM>
constexpr int kNnCoarseMax = nn_coarse_for((16 * NM - kConstSlots) / kPieceGroups);
template <int NM, int TQ>
__global__ __launch_bounds__(256, 2) void nn_mfma_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const uint4* __restrict__ img_b, const float* __restrict__ norms,
    const uint4* __restrict__ img_s, const float* __restrict__ norms_s,
    const uint32_t* __restrict__ perm, const uint32_t* __restrict__ invpos,
    const uint32_t* __restrict__ pq_of, const uint32_t* __restrict__ hdr, uint32_t T,
    uint32_t i_from, uint32_t i_to, uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
    uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  if (hdr[1] != 0) return;
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t qt0 = i_from / 32 + wave * TQ;
  if (qt0 * 32 >= i_to) return;

  // (scaled units, like the accumulators and the running minima taken from them)
  const Scale sc = load_scale(hdr);   // (the neighbour scale: scale_kernel ran before the images were built)
  const GuardBand gb = guard_band(__uint_as_float(hdr[0]) * sc.s2, 0.0f, (int)n_cols, sc);

  s16x8 b[TQ][NM];
  NnQ q[TQ];
  uint32_t jq[TQ];
  uint64_t livemask[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    jq[qt] = tile * 32 + c;
    const bool live = (tile < T) && (jq[qt] >= i_from) && (jq[qt] < i_to);
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    const uint32_t tl = tile < T ? tile : T - 1;
    load_query<NM>(img_b, tl, lane, h, live ? norms[tl * 32 + c] : dead_const(sc), sc, b[qt]);
    const uint32_t jl = live ? jq[qt] : (n_rows - 1);
    q[qt].pq = live ? pq_of[jl] : 0u;
    q[qt].spos = live ? invpos[jl] : 0xFFFFFFFFu;
    q[qt].t_self = live ? (q[qt].spos >> 5) : 0xFFFFFFFFu;
    q[qt].t_full = q[qt].pq >> 5;
    q[qt].t_part = (q[qt].pq & 31u) ? (q[qt].pq >> 5) : 0xFFFFFFFFu;
    // idle lanes start at -inf: they can never trigger the exact path and never change
    q[qt].m_nn = live ? INFINITY : -INFINITY;
    q[qt].m_hd = live ? INFINITY : -INFINITY;
    q[qt].bd_nn = FLT_MAX;
    q[qt].bd_hd = FLT_MAX;
    q[qt].bj_nn = n_rows + 1;
    q[qt].bj_hd = n_rows + 1;
  }

  s16x8 a0[NM], a1[NM];
  float4 n0[4], n1[4];
  load_tile<NM>(img_s, norms_s, 0, lane, h, a0, n0);

  // the rest of an epilogue: minima of the special tiles, band test against the running minima, rare
  // exact path.  Bitwise logic on purpose (no short-circuit control flow in the hot path).
  auto finish = [&](const f32x16& acc, auto qi_c, float tmin, uint32_t t) {
    constexpr int qi = decltype(qi_c)::value;
    NnQ& Q = q[qi];
    const bool special = (t == Q.t_self) | (t == Q.t_part);
    float hmin = (t < Q.t_full) ? tmin : INFINITY;
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(special) != 0, 0)) {
      const NnMin g = nn_special(acc, t, h, Q.spos, Q.pq);   // valid for every lane, just slower
      tmin = g.tmin;
      hmin = g.hmin;
    }
    const float new_nn = fminf(Q.m_nn, tmin), new_hd = fminf(Q.m_hd, hmin);
    const float bn = nn_band(gb, new_nn), bh = nn_band(gb, new_hd);
    const bool trig = (tmin < bn) | (hmin < bh);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(trig) != 0, 0)) {
      const bool live = (livemask[qi] >> lane) & 1;
      NnBest best{Q.bd_nn, Q.bd_hd, Q.bj_nn, Q.bj_hd};
      best = nn_fix(coords, perm, n_rows, n_cols, acc, bn, bh, best, jq[qi], Q.spos, Q.pq, t, h);
      Q.bd_nn = live ? best.bd_nn : Q.bd_nn;
      Q.bj_nn = live ? best.bj_nn : Q.bj_nn;
      Q.bd_hd = live ? best.bd_hd : Q.bd_hd;
      Q.bj_hd = live ? best.bj_hd : Q.bj_hd;
    }
    Q.m_nn = new_nn;
    Q.m_hd = new_hd;
  };

  // chains software-pipelined over two accumulator tiles (see nn_pruned_kernel)
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
  uint32_t tB = 0;
  auto tile_body = [&](const s16x8 (&a)[NM], const float4 (&nv)[4], uint32_t t) {
    const f32x16 c0 = frag16(nv);
    if constexpr (TQ == 1) {
      float tmin = INFINITY;
      nn_chain<NM>(a, b[0], c0, accA, accB, tmin);
      finish(accB, std::integral_constant<int, 0>{}, tmin, tB);
      accB = accA;
    } else {
      constexpr_for_pairs<TQ>([&](auto qt_c) {
        constexpr int qt = decltype(qt_c)::value;
        constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
        float tmin = INFINITY;
        nn_chain<NM>(a, b[qt], c0, accA, accB, tmin);
        finish(accB, std::integral_constant<int, qb>{}, tmin, (qt == 0) ? tB : t);
        tmin = INFINITY;
        nn_chain<NM>(a, b[qt + 1], c0, accB, accA, tmin);
        finish(accA, std::integral_constant<int, qt>{}, tmin, t);
      });
    }
    keep_alive(c0);
    tB = t;
  };

  for (uint32_t t = 0; t < T; t += 2) {
    load_tile<NM>(img_s, norms_s, (t + 1 < T) ? t + 1 : t, lane, h, a1, n1);
    tile_body(a0, n0, t);
    if (t + 1 < T) {
      load_tile<NM>(img_s, norms_s, (t + 2 < T) ? t + 2 : t + 1, lane, h, a0, n0);
      tile_body(a1, n1, t + 1);
    }
  }
  {  // drain: epilogue of the last pending chain
    float tmin = INFINITY;
    tile_min<0, 16>(accB, tmin);
    finish(accB, std::integral_constant<int, TQ - 1>{}, tmin, tB);
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    NnQ& Q = q[qt];
    // merge the two half-waves (disjoint reference rows of the same query)
    float od = __shfl_xor(Q.bd_nn, 32, 64);
    uint32_t oj = (uint32_t)__shfl_xor((int)Q.bj_nn, 32, 64);
    lexi_update(oj <= n_rows, Q.bd_nn, Q.bj_nn, od, oj, n_rows);
    od = __shfl_xor(Q.bd_hd, 32, 64);
    oj = (uint32_t)__shfl_xor((int)Q.bj_hd, 32, 64);
    lexi_update(oj <= n_rows, Q.bd_hd, Q.bj_hd, od, oj, n_rows);
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      nn_idx[jq[qt]] = Q.bj_nn;
      nn_d2[jq[qt]] = Q.bd_nn;
      hd_idx[jq[qt]] = Q.bj_hd;
      hd_d2[jq[qt]] = Q.bd_hd;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// neighbour sweep over frames ordered by (2-D cell, free energy) with ring-wise tile pruning.
//
// Order: cells of a coarse 2-D grid on columns 0/1 (row-major), frames of a cell by ascending free
// energy.  A tile therefore is (a) compact in the (col 0, col 1) plane -> bounding box, and (b) has
// a narrow free-energy range [fe_lo, fe_hi] -> for a query with free energy feq the tile is
// entirely lower (fe_hi < feq: hd minimum = nn minimum), entirely not lower (fe_lo >= feq) or mixed
// (masked per-element epilogue, like the tile that holds the query itself).
//
// Pruning: a wave processes reference tiles in rings of growing box distance from its query group.
// After a ring with outer radius R every unvisited frame is at least R away (d2 in D dimensions >=
// box gap in two of them), so a query whose EXACT incumbent d2 is < R^2 is settled.  The next radius
// is the largest incumbent still to be confirmed (or 4x, while some query has no candidate yet).
// Typical: one local ring, one confirming ring, done -- far clusters are never touched.
// ---------------------------------------------------------------------------------------------
struct NnPQ {            // per query tile, per lane
  float feq;             // free energy of the query
  float m_nn, m_hd;      // running minima of the MFMA values over the visited reference rows
  float bn, bh;          // nn_band(m_nn), nn_band(m_hd): candidates are the values below these
  float bd_nn, bd_hd;    // exact incumbents (canonical d2)
  uint32_t bj_nn, bj_hd;
  uint32_t spos;         // position of the query itself in the reference order
};

// the same without the exact incumbents (nn_pruned_kernel keeps them in LDS only: best64)
struct NnPQr {
  float feq, m_nn, m_hd, bn, bh;
  uint32_t spos;
};

__device__ __attribute__((noinline)) NnMin nn_special_fe(f32x16 acc, const float* __restrict__ fe_c,
                                                         uint32_t t, int h, uint32_t spos,
                                                         float feq) {
  NnMin o{INFINITY, INFINITY};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t pos = tile_row(t, r, h);
    const float v = (pos != spos) ? acc[r] : INFINITY;
    o.tmin = fminf(o.tmin, v);
    o.hmin = fminf(o.hmin, (fe_c[pos] < feq) ? v : INFINITY);   // fe_c is +inf padded
  }
  return o;
}

// The exact path reads the query row from LDS (staged once per wave) and the reference row from a
// copy of the ORIGINAL coordinates gathered into the reference order: one global-load latency per
// call instead of two dependent ones (permutation, then row).
__device__ __attribute__((noinline)) NnBest nn_fix_fe(const float* __restrict__ coords_c,
                                                      const uint32_t* __restrict__ perm,
                                                      const float* qrow, uint32_t n_rows,
                                                      uint32_t n_cols, f32x16 acc, f32x16 fef,
                                                      float bn, float bh, NnBest best,
                                                      uint32_t spos, float feq, uint32_t t, int h) {
  // 1. which of this lane's 16 elements are candidates (no memory traffic)
  uint32_t mn = 0, mh = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t pos = tile_row(t, r, h);
    const bool other = (pos != spos) & (pos < n_rows);
    mn |= (other & (acc[r] < bn)) ? (1u << r) : 0u;
    mh |= (other & (acc[r] < bh) & (fef[r] < feq)) ? (1u << r) : 0u;
  }
  // 2. every lane evaluates ITS next candidate in the same iteration: the row fetches of all lanes
  //    overlap, the loop runs max-candidates-per-lane times (usually once)
  uint32_t m = mn | mh;
  while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
    if (m != 0) {
      const int r = __builtin_ctz(m);
      const uint32_t pos = tile_row(t, r, h);
      const uint32_t j = perm[pos];
      const float d2c = dist2_canon_rows(qrow, coords_c + (size_t)pos * n_cols, (int)n_cols);
      lexi_update((mn >> r) & 1u, best.bd_nn, best.bj_nn, d2c, j, n_rows);
      lexi_update((mh >> r) & 1u, best.bd_hd, best.bj_hd, d2c, j, n_rows);
      m &= m - 1;
    }
  }
  return best;
}

// ---- deferred exact evaluation -------------------------------------------------------------------
// The neighbour sweep does not evaluate a candidate the moment it appears (one or two lanes busy, a
// full memory latency each time); it parks (position, kind) in a small per-lane LDS queue and
// evaluates them in batches: every lane of the wave then fetches and evaluates ONE entry
// in parallel.  The approximate running minima that select candidates never depend on exact
// values, so deferring changes nothing but the order of the lexicographic merges.
constexpr uint32_t kQueuePosMask = 0x3FFFFFFFu;   // entry = position | nn-flag << 30 | hd-flag << 31

// Wave-wide form (the pruned neighbour sweep): candidates of all lanes go into ONE compact list (x = position |
// nn-flag << 30 | hd-flag << 31, y = query tile * 32 + column) and are evaluated 64 at a time, one per lane.
// The exact incumbents of a query live in LDS as order-preserving words (d2 bits << 32 | frame id): the lexicographic
// merge "strict < on d2 scanning ascending j" is a 64-bit unsigned minimum (d2 >= 0; the initial (FLT_MAX, n_rows + 1)
// is the largest word a real candidate can be compared with), shared by the two half-wave lanes of the query.
__device__ __attribute__((noinline)) void nn_wave_flush(const uint2* queue, uint32_t qn, const float* qrows,
                                                        unsigned long long* best /* [2][n_queries] */,
                                                        uint32_t n_queries, const float* __restrict__ coords_c,
                                                        const uint32_t* __restrict__ perm, uint32_t n_cols,
                                                        int lane) {
  for (uint32_t k0 = 0; k0 < qn; k0 += 64) {
    if (k0 + lane < qn) {
      const uint2 ent = queue[k0 + lane];
      const uint32_t pos = ent.x & kQueuePosMask, qidx = ent.y;
      const uint32_t j = perm[pos];
      const float d2c = dist2_canon_rows(qrows + (size_t)qidx * n_cols, coords_c + (size_t)pos * n_cols, (int)n_cols);
      const unsigned long long key = ((unsigned long long)__float_as_uint(d2c) << 32) | j;
      if ((ent.x >> 30) & 1u) atomicMin(&best[qidx], key);
      if ((ent.x >> 31) & 1u) atomicMin(&best[n_queries + qidx], key);
    }
  }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---- reference norms folded into the operands (nn_pruned_kernel) --------------------------------------------------
// The pruned neighbour sweep takes the REFERENCE row's norm through the two constant slots (A side: the pieces of
// |y'|^2 / 2^a, image_kernel mode 2; B side: 2^a twice, patched in here) and starts every chain from C = 0, so a
// reference tile is NM x 16 B per lane and nothing else: no 4 x dwordx4 of norms per tile, no 16 + 16 registers for
// them.  The accumulator is then acc' = |y'|^2 - 2 x'.y' = d2 - |x'|^2: the QUERY norm c_q stays outside, a per-lane
// constant that moves into the thresholds (a threshold T on d2 is the threshold T - c_q on acc', rounded up: only
// ever more candidates).  Pad reference rows carry 65504 * 2^a (~2^31) in slot 0 and nothing else; thresholds are
// capped at kNnThrCap = 1.25 * 2^30 before c_q comes off (every real acc' + c_q is a d2 <= 4 S M < 2^30), so a pad
// row is never a candidate, also while a query has no incumbent yet (threshold +inf).  What the band pays: the
// accumulator no longer collapses to ~d2 after the hi x hi products, the ns later MFMAs truncate their addends at
// 2^-24 of ~M instead of ~d2: + 18 ns u M in e0 (guard_e0_linear, `folded`).
constexpr float kNnThrCap = 1342177280.0f;   // 1.25 * 2^30
__device__ __forceinline__ float round_up(float x) {   // >= x for finite x (one ulp or so); +-inf unchanged
  return (fabsf(x) < INFINITY) ? x + fabsf(x) * 1.1920929e-7f : x;
}
__device__ __forceinline__ float nn_prime(float thr, float cq) { return round_up(fminf(thr, kNnThrCap) - cq); }   // d2 units -> acc' units
__device__ __forceinline__ float nn_unprime(float v, float cq) { return round_up(v + cq); }                       // acc' units -> d2 units
template <int NM>
__device__ __forceinline__ void load_query_folded(const uint4* __restrict__ img_b, uint32_t tile, int lane, int h,
                                                  bool live, const Scale& sc, s16x8 (&b)[NM]) {
  const uint4* ip = img_b + (size_t)tile * (NM * 64) + lane;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const uint4 v = ip[m * 64];
    b[m] = __builtin_bit_cast(s16x8, v);
  }
  if (h == 0) {
    const short one = live ? (short)const_a_bits(sc.a) : (short)0;
    b[0][0] = one;
    b[0][1] = one;
  }
}
template <int NM>
__device__ __forceinline__ void load_tile_folded(const uint4* __restrict__ img, uint32_t t, int lane, s16x8 (&a)[NM]) {
  const uint4* ip = img + (size_t)t * (NM * 64) + lane;
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const uint4 v = ip[m * 64];
    a[m] = __builtin_bit_cast(s16x8, v);
  }
}

// COOP (round 6): the waves of a workgroup are consecutive reference SHARES of ONE query group instead of different
// groups.  They keep one copy of the query rows, of the exact incumbents (best64) and of the published bounds in LDS,
// and they exchange the running minima that set the candidate thresholds through LDS inside the candidate path (smin:
// a returning ds_min per query and kind, only in the chains that went on): a threshold one share has learnt is learnt
// for all of them -- DESIGN.md 4.8 measured 0.45 ms of a rank's 1.6 ms (an eighth of C3) as thresholds learnt again by
// every one of a group's 34 shares.  Taken when a group has many shares (sharded runs); a share is still a wave.
template <int NM, int TQ, bool COOP = false>
__global__ __launch_bounds__(COOP ? 512 : 256, COOP ? 1 : 2) void nn_pruned_kernel(
    const float* __restrict__ coords, uint32_t n_rows, uint32_t n_cols,
    const float* __restrict__ fe, const uint4* __restrict__ img_r,
    const float* __restrict__ norms_r, const uint32_t* __restrict__ perm_r,
    const float4* __restrict__ box_r, const float4* __restrict__ box_t,
    const float2* __restrict__ ferange_r,
    const float* __restrict__ fe_c, const float* __restrict__ coords_c,
    const uint32_t* __restrict__ invpos_r, uint32_t T,
    const uint4* __restrict__ img_q, const float* __restrict__ norms_q,
    const uint32_t* __restrict__ perm_q, const float4* __restrict__ box_q, uint32_t n_q, QSeg q_seg,
    int full_range, float cell2,
    const uint32_t* __restrict__ hdr, unsigned long long* __restrict__ chain_counter,
    unsigned long long* __restrict__ merge64, uint32_t* __restrict__ nn_idx,
    float* __restrict__ nn_d2, uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2, CompView CV) {
  // dynamic LDS, per wave of the workgroup: the survivor list of a scan round [kListCap], then [TQ*32][n_cols] query
  // rows (original coordinates), then the candidate queues [TQ][kQueueCap][64]
  extern __shared__ __attribute__((aligned(16))) float nn_dyn_lds[];
  if (hdr[1] != 0) return;
  const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
  const int wib = threadIdx.x >> 6;
  // waves per workgroup (kWavesPerGroup): the waves share nothing, and a workgroup holds its LDS until its LAST
  // wave is done -- with four waves of unequal cost a fifth of the wave slots sat idle (SQ_WAVE_CYCLES: 1.57 of 2
  // per SIMD at C3), with one wave per workgroup a finished wave is replaced at once (C3: 16.3 -> 15.2 ms)
  const uint32_t wpb = blockDim.x >> 6;
  uint32_t* lists_all = reinterpret_cast<uint32_t*>(nn_dyn_lds);
  float* qrows_all = nn_dyn_lds + (size_t)wpb * kListCap;
  const uint32_t TQT = (n_q + 31) / 32;
  const uint32_t blk_unit = xcd_block(COOP ? seg_groups((TQT + TQ - 1) / TQ, q_seg) : (seg_groups((TQT + TQ - 1) / TQ, q_seg) + wpb - 1) / wpb);
  if (blk_unit == 0xFFFFFFFFu) return;   // (pad block of the grid)
  const uint32_t wave = seg_group(COOP ? blk_unit : blk_unit * wpb + wib, q_seg);
  // reference tiles dealt round-robin (COOP: the workgroup's waves take neighbouring shares)
  const uint32_t chunk = COOP ? blockIdx.y * wpb + (uint32_t)wib : blockIdx.y, n_chunks = COOP ? gridDim.y * wpb : gridDim.y;
  const uint32_t qt0 = wave * TQ;
  if (qt0 >= TQT) return;   // (COOP: the whole workgroup leaves -- no barrier is ever met by a part of it)
  // (the component of the group, its tile range and cell edge: three dependent look-ups, started before everything else)
  const uint32_t my_comp = CV.tile_comp_q[qt0];
  const uint32_t comp_lo = CV.range_r[2 * my_comp], comp_hi = CV.range_r[2 * my_comp + 1];
  const float comp_cell = __uint_as_float(CV.comp[kCompFine + 4 * min(my_comp, (uint32_t)kMaxComp - 1u) + 2]);
  uint32_t* list = lists_all + (size_t)wib * kListCap;
  // COOP layout (fixed offsets first): lists [wpb][kListCap] | candidate lists [wpb][2 kWaveQueue] | best64 [2][TQ*32] x 8 B |
  // g_pub [2][TQ*32] | smin [2][TQ*32] | query rows [TQ*32][n_cols] -- everything behind the lists once per WORKGROUP
  uint32_t* coop_base = lists_all + (size_t)wpb * kListCap;
  float* qrows = COOP ? reinterpret_cast<float*>(coop_base + (size_t)wpb * 2 * kWaveQueue + 8 * TQ * 32)
                      : qrows_all + (size_t)wib * (TQ * 32) * n_cols;
  uint32_t* queues = COOP ? coop_base + (size_t)wib * 2 * kWaveQueue
                          : reinterpret_cast<uint32_t*>(qrows_all + (size_t)wpb * (TQ * 32) * n_cols) +
                                (size_t)wib * (TQ * kQueueCap * 64 + 2 * TQ * 32);
  // the wave's LDS behind the query rows (TQ * kQueueCap * 64 words): the compact candidate list (kWaveQueue
  // entries of 8 B) and the packed exact incumbents [2][TQ*32] of 8 B
  static_assert(TQ * kQueueCap * 64 >= 2 * kWaveQueue + 4 * TQ * 32, "candidate list + incumbents fit the queue region");
  uint2* cand = reinterpret_cast<uint2*>(queues);
  unsigned long long* best64 = reinterpret_cast<unsigned long long*>(COOP ? coop_base + (size_t)wpb * 2 * kWaveQueue : queues + 2 * kWaveQueue);
  // exact incumbents other reference chunks had published when this wave started (FLT_MAX: none), [2][TQ*32] -- kept in
  // LDS like the wave's own incumbents: neither is touched inside the chains, and as registers they were live (and
  // partly spilled) across the whole sweep
  float* g_pub = COOP ? reinterpret_cast<float*>(coop_base + (size_t)wpb * 2 * kWaveQueue + 4 * TQ * 32)
                      : reinterpret_cast<float*>(queues + TQ * kQueueCap * 64);   // (2 * TQ * 32 words behind the queue region)
  // COOP: the group's running minima (ordered-integer keys of m_nn / m_hd in d2 units), [2][TQ*32]
  uint32_t* smin = coop_base + (size_t)wpb * 2 * kWaveQueue + 6 * TQ * 32;
  uint32_t qn = 0;   // queued candidates (wave-uniform)

  // (scaled units, like the accumulators and the running minima taken from them)
  const Scale sc = load_scale(hdr);   // (the neighbour scale: scale_kernel ran before the images were built)
  const GuardBand gb = guard_band(__uint_as_float(hdr[kHdrMused]) * sc.s2, 0.0f, (int)n_cols, sc, true);   // (the extent the scale was chosen for; folded norms)
  (void)cell2;   // (the first ring's floor: the cell edge of the query's own component, set below)
  (void)norms_r;   // (the reference norms ride in the operand image)
  // (early-out of the chains, see nn_chain_coarse: the cached thresholds q[].bn / q[].bh INCLUDE the skip bound -- the
  //  coarse test compares against them as they are, and the test in front of the candidate path, which only has to let
  //  every candidate through, uses them too; the exact bands are formed again inside that path)
  const float skipb = kNnEarly<NM> ? nn_skip_bound(__uint_as_float(hdr[kHdrMused]) * sc.s2) : 0.0f;

  s16x8 b[TQ][NM];
  NnPQr q[TQ];   // (m_nn, m_hd in d2 units; bn, bh in the accumulators' units: c_q taken off)
  float cq[TQ];  // |x'|^2 of the lane's query (scaled units)
  uint32_t jq[TQ];
  uint64_t livemask[TQ];
  float4 qbox[TQ];
  float4 gbox = make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
  float g_nn[TQ], g_hd[TQ];
#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    const uint32_t tile = qt0 + qt;
    const uint32_t tl = tile < TQT ? tile : TQT - 1;
    const uint32_t pos = tile * 32 + c;
    const uint32_t frame = ((tile < TQT) && (pos < n_q)) ? perm_q[pos] : kInvalidFrame;   // (pad positions: kInvalidFrame)
    const bool live = frame != kInvalidFrame;
    livemask[qt] = __builtin_amdgcn_ballot_w64(live);
    jq[qt] = live ? frame : 0u;
    load_query_folded<NM>(img_q, tl, lane, h, live, sc, b[qt]);
    cq[qt] = live ? norms_q[tl * 32 + c] : 0.0f;
    q[qt].feq = live ? fe[jq[qt]] : -INFINITY;
    q[qt].spos = live ? (full_range ? pos : invpos_r[jq[qt]]) : 0xFFFFFFFFu;
    q[qt].m_nn = live ? INFINITY : -INFINITY;   // idle lanes can never trigger the exact path
    q[qt].m_hd = live ? INFINITY : -INFINITY;
    // what the waves of other reference chunks have already published for this query: an exact upper bound.  Only
    // candidates that can still beat (or tie) it need to be looked at, i.e. MFMA values below d2 + eps(d2); the band
    // test adds its usual margin on top.  (Branch-free, and nothing goes to LDS inside this loop: the loads of the four
    // tiles are issued together -- with a store per tile in between the set-up of a wave was 32 000 cycles, an eighth of
    // a wave of an eight-way sharded sweep.)
    const bool pub = (n_chunks > 1) & live;
    const unsigned long long w_nn = merge64[pub ? jq[qt] : 0u], w_hd = merge64[pub ? (size_t)n_rows + jq[qt] : 0u];
    g_nn[qt] = pub ? __uint_as_float((uint32_t)(w_nn >> 32)) : FLT_MAX;
    g_hd[qt] = pub ? __uint_as_float((uint32_t)(w_hd >> 32)) : FLT_MAX;
    {
      const float s_nn = g_nn[qt] * sc.s2, s_hd = g_hd[qt] * sc.s2;   // (exact d2 -> scaled units)
      if (g_nn[qt] < FLT_MAX) q[qt].m_nn = s_nn + (gb.e0 + gb.kappa * s_nn);
      if (g_hd[qt] < FLT_MAX) q[qt].m_hd = s_hd + (gb.e0 + gb.kappa * s_hd);
    }
    q[qt].m_nn = fminf(q[qt].m_nn, q[qt].m_hd);   // (two reads of merge64 a moment apart: keep m_nn <= m_hd)
    q[qt].bn = nn_prime(nn_band(gb, q[qt].m_nn), cq[qt]) + skipb;
    q[qt].bh = nn_prime(nn_band(gb, q[qt].m_hd), cq[qt]) + skipb;
    qbox[qt] = (tile < TQT) ? box_q[tile] : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
    gbox.x = fminf(gbox.x, qbox[qt].x);
    gbox.y = fmaxf(gbox.y, qbox[qt].y);
    gbox.z = fminf(gbox.z, qbox[qt].z);
    gbox.w = fmaxf(gbox.w, qbox[qt].w);
  }
  if constexpr (COOP) {
    // one copy per workgroup: published bounds, exact incumbents, running minima (wave 0), query rows (all waves)
    if (wib == 0 && h == 0) {
#pragma unroll
      for (int qt = 0; qt < TQ; ++qt) {
        g_pub[qt * 32 + c] = g_nn[qt];
        g_pub[TQ * 32 + qt * 32 + c] = g_hd[qt];
        best64[qt * 32 + c] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | (n_rows + 1);
        best64[TQ * 32 + qt * 32 + c] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | (n_rows + 1);
        smin[qt * 32 + c] = fkey(q[qt].m_nn);
        smin[TQ * 32 + qt * 32 + c] = fkey(q[qt].m_hd);
      }
    }
    if (full_range && qt0 + TQ <= TQT) {
      const float* src = coords_c + (size_t)qt0 * 32 * n_cols;
      const uint32_t total = (uint32_t)TQ * 32u * n_cols, nthr = blockDim.x;
      for (uint32_t e0 = threadIdx.x; e0 < total; e0 += 8u * nthr) {
        float v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) v[j] = (e0 + nthr * j < total) ? src[e0 + nthr * j] : 0.0f;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
          if (e0 + nthr * j < total) qrows[e0 + nthr * j] = v[j];
      }
    } else {
#pragma unroll
      for (int qt = 0; qt < TQ; ++qt) {
        if (((uint32_t)qt % wpb) != (uint32_t)wib) continue;
        const uint32_t tile = qt0 + qt, tl = tile < TQT ? tile : TQT - 1;
        stage_query_rows(qrows + (size_t)qt * 32 * n_cols, full_range ? coords_c + (size_t)tl * 32 * n_cols : nullptr, coords,
                         jq[qt], (livemask[qt] >> lane) & 1, n_cols, lane);
      }
    }
    __syncthreads();
    if (blockIdx.y == 0) {
      // Seeds (see below), dealt to the waves of the group's FIRST workgroup: wave w looks at the frames w + 1, w + 1 + wpb,
      // ... positions away; the results meet in LDS and are published at once
#pragma unroll
      for (int qt = 0; qt < TQ; ++qt) {
        const bool live = (livemask[qt] >> lane) & 1;
        float bd_nn = FLT_MAX, bd_hd = FLT_MAX;
        uint32_t bj_nn = n_rows + 1, bj_hd = n_rows + 1;
        if (live) {
          const float* qrow = qrows + (qt * 32 + c) * n_cols;
          for (int k = 1 + wib; k <= kSeedNeighbours; k += (int)wpb) {
            const long long p2 = (long long)q[qt].spos + (h ? -k : k);
            if (p2 >= 0 && p2 < (long long)CV.n_pos && perm_r[p2] != kInvalidFrame) {
              const float d2c = dist2_canon_rt(qrow, 1, coords_c + (size_t)p2 * n_cols, 1, (int)n_cols);
              const uint32_t j = perm_r[p2];
              lexi_update(true, bd_nn, bj_nn, d2c, j, n_rows);
              lexi_update(fe_c[p2] < q[qt].feq, bd_hd, bj_hd, d2c, j, n_rows);
            }
          }
          const float s_nn = bd_nn * sc.s2, s_hd = bd_hd * sc.s2;
          const unsigned long long w_nn = ((unsigned long long)__float_as_uint(bd_nn) << 32) | bj_nn;
          const unsigned long long w_hd = ((unsigned long long)__float_as_uint(bd_hd) << 32) | bj_hd;
          if (bd_nn < FLT_MAX) {
            atomicMin(&best64[qt * 32 + c], w_nn);
            atomicMin(&smin[qt * 32 + c], fkey(s_nn + (gb.e0 + gb.kappa * s_nn)));
            atomicMin(&merge64[jq[qt]], w_nn);
          }
          if (bd_hd < FLT_MAX) {
            atomicMin(&best64[TQ * 32 + qt * 32 + c], w_hd);
            atomicMin(&smin[TQ * 32 + qt * 32 + c], fkey(s_hd + (gb.e0 + gb.kappa * s_hd)));
            atomicMin(&merge64[(size_t)n_rows + jq[qt]], w_hd);
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      NnPQr& Q = q[qt];
      if ((livemask[qt] >> lane) & 1) {
        Q.m_hd = fminf(Q.m_hd, fkey_inv(smin[TQ * 32 + qt * 32 + c]));
        Q.m_nn = fminf(fminf(Q.m_nn, fkey_inv(smin[qt * 32 + c])), Q.m_hd);
        Q.bn = nn_prime(nn_band(gb, Q.m_nn), cq[qt]) + skipb;
        Q.bh = nn_prime(nn_band(gb, Q.m_hd), cq[qt]) + skipb;
      }
    }
  } else {
    // published bounds and the original coordinates of the queries (for the exact path) into LDS
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt)
      if (h == 0) {
        g_pub[qt * 32 + c] = g_nn[qt];
        g_pub[TQ * 32 + qt * 32 + c] = g_hd[qt];
      }
    if (full_range && qt0 + TQ <= TQT) {
      // (the rows of the wave's tiles are one contiguous piece of the ordered copy: twenty loads per lane in flight, all
      //  of them at ten columns)
      const float* src = coords_c + (size_t)qt0 * 32 * n_cols;
      const uint32_t total = (uint32_t)TQ * 32u * n_cols;
      for (uint32_t e0 = (uint32_t)lane; e0 < total; e0 += 1280u) {
        float v[20];
#pragma unroll
        for (uint32_t j = 0; j < 20; ++j) v[j] = (e0 + 64u * j < total) ? src[e0 + 64u * j] : 0.0f;
#pragma unroll
        for (uint32_t j = 0; j < 20; ++j)
          if (e0 + 64u * j < total) qrows[e0 + 64u * j] = v[j];
      }
    } else {
#pragma unroll
      for (int qt = 0; qt < TQ; ++qt) {
        const uint32_t tile = qt0 + qt, tl = tile < TQT ? tile : TQT - 1;
        stage_query_rows(qrows + (size_t)qt * 32 * n_cols, full_range ? coords_c + (size_t)tl * 32 * n_cols : nullptr, coords,
                         jq[qt], (livemask[qt] >> lane) & 1, n_cols, lane);
      }
    }
    // Seeds: the frames next to each query in the sweep's order (same cell, neighbouring free energy; the
    // ones before it have a lower free energy) are evaluated exactly before the first ring.  They are
    // ordinary candidates; what they buy is finite running minima from the start -- without them the first
    // tiles of a sweep park every element, and query groups in sparse regions (wide boxes, long first ring)
    // spent a microsecond per chain in the candidate path.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // (query rows: written by the h = 0 lanes)
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      const bool live = (livemask[qt] >> lane) & 1;
      NnPQr& Q = q[qt];
      float bd_nn = FLT_MAX, bd_hd = FLT_MAX;          // exact incumbents of this query (canonical d2, frame id)
      uint32_t bj_nn = n_rows + 1, bj_hd = n_rows + 1;
      // (only the first reference share: the later ones start from what the earlier ones published)
      if (live && chunk == 0) {
        const float* qrow = qrows + (qt * 32 + c) * n_cols;
        for (int k = 1; k <= kSeedNeighbours; ++k) {
          const long long p2 = (long long)Q.spos + (h ? -k : k);
          if (p2 >= 0 && p2 < (long long)CV.n_pos && perm_r[p2] != kInvalidFrame) {
            const float d2c = dist2_canon_rt(qrow, 1, coords_c + (size_t)p2 * n_cols, 1, (int)n_cols);
            const uint32_t j = perm_r[p2];
            lexi_update(true, bd_nn, bj_nn, d2c, j, n_rows);
            lexi_update(fe_c[p2] < Q.feq, bd_hd, bj_hd, d2c, j, n_rows);
          }
        }
        // (the two half-wave lanes looked at the frames after / before the query: merge, then both hold the result)
        {
          float od = __shfl_xor(bd_nn, 32, 64);
          uint32_t oj = (uint32_t)__shfl_xor((int)bj_nn, 32, 64);
          lexi_update(oj <= n_rows, bd_nn, bj_nn, od, oj, n_rows);
          od = __shfl_xor(bd_hd, 32, 64);
          oj = (uint32_t)__shfl_xor((int)bj_hd, 32, 64);
          lexi_update(oj <= n_rows, bd_hd, bj_hd, od, oj, n_rows);
        }
        const float s_nn = bd_nn * sc.s2, s_hd = bd_hd * sc.s2;
        if (bd_nn < FLT_MAX) Q.m_nn = fminf(Q.m_nn, s_nn + (gb.e0 + gb.kappa * s_nn));
        if (bd_hd < FLT_MAX) Q.m_hd = fminf(Q.m_hd, s_hd + (gb.e0 + gb.kappa * s_hd));
        Q.bn = nn_prime(nn_band(gb, Q.m_nn), cq[qt]) + skipb;
        Q.bh = nn_prime(nn_band(gb, Q.m_hd), cq[qt]) + skipb;
        // published at once: the other shares of this group start while this wave is still sweeping
        if (n_chunks > 1) {
          if (bd_nn < FLT_MAX)
            atomicMin(&merge64[jq[qt]], ((unsigned long long)__float_as_uint(bd_nn) << 32) | bj_nn);
          if (bd_hd < FLT_MAX)
            atomicMin(&merge64[(size_t)n_rows + jq[qt]],
                      ((unsigned long long)__float_as_uint(bd_hd) << 32) | bj_hd);
        }
      }
      // the exact incumbents of the wave's queries live in LDS as order-preserving words (see nn_wave_flush)
      if (h == 0) {
        best64[qt * 32 + c] = ((unsigned long long)__float_as_uint(bd_nn) << 32) | bj_nn;
        best64[TQ * 32 + qt * 32 + c] = ((unsigned long long)__float_as_uint(bd_hd) << 32) | bj_hd;
      }
    }
  }
  // lowest free energy of the whole data set (header word 12, ordered-integer key, written by the
  // ordering pass): a query at that level has no lower-FE neighbour
  const float fe_floor = fkey_inv(~hdr[12]);
  // evaluate and empty the candidate list (64 candidates at a time, one per lane)
  auto flush = [&]() {
    nn_wave_flush(cand, qn, qrows, best64, TQ * 32, coords_c, perm_r, n_cols, lane);
    qn = 0;
  };
  uint32_t chains = 0, visited = 0;
  uint32_t chains_on = 0;   // chains that went on behind the early-out test (computed again in full)
  // this wave's share of the reference tiles: t = chunk + u * n_chunks, u = 0 .. U-1 (round-robin, so
  // every share sees every region; the scans only touch their own boxes)
  // (the tiles of the group's own COMPONENT only -- dc_mfma_kernels.hpp "components": what lies in other components is
  //  looked at afterwards, exactly, for the few queries whose neighbours may be there: nn_cross_kernel)
  const uint32_t t_lo = comp_lo, t_hi = min(comp_hi, T);
  const uint32_t u_lo = (t_lo > chunk) ? (t_lo - chunk + n_chunks - 1) / n_chunks : 0u;
  const uint32_t U = max((t_hi > chunk) ? (t_hi - chunk + n_chunks - 1) / n_chunks : 0u, u_lo);
  cell2 = comp_cell * comp_cell;
  const uint32_t U_stride = (T + n_chunks - 1) / n_chunks;   // boxes of a share in box_t
  const float dgx = gbox.y - gbox.x, dgy = gbox.w - gbox.z;
  float r2_lo = -1.0f;                                     // rings: r2_lo <= gap2 < r2_hi
  float r2_hi = fmaxf(dgx * dgx + dgy * dgy, cell2);
  if (!(r2_hi > 0.0f)) r2_hi = FLT_MIN;
  for (;;) {
    for (uint32_t base = u_lo; base < U; base += kListCap) {
      uint32_t cnt = 0;
      const uint32_t lim = min(U - base, (uint32_t)kListCap);
      auto tile_of = [&](uint32_t u) { return chunk + u * n_chunks; };
      // (box_t: the boxes of this share stored contiguously -- a step of the scan reads 1 KB instead of
      //  gathering 64 cache lines gridDim.y tiles apart)
      const float4* box_s = box_t + (size_t)chunk * U_stride;
      float4 rb_next = ((uint32_t)lane < lim) ? box_s[base + lane]
                                              : make_float4(INFINITY, -INFINITY, INFINITY, -INFINITY);
      for (uint32_t k = 0; k < lim; k += 64) {
        const uint32_t t = tile_of(base + k + lane);
        const float4 rb = rb_next;   // fetched one step ahead: the scan is latency-bound otherwise
        if (k + 64 + lane < lim) rb_next = box_s[base + k + 64 + lane];
        bool ok = false;
        if (k + lane < lim) {
          const float g2 = box_gap2(gbox, rb);
          ok = (g2 < r2_hi) & (g2 >= r2_lo);
        }
        // (the ring logic below only ever sees this wave's share of the references: its incumbents
        //  are upper bounds of the true ones, so the rings are at worst a little wider than necessary)
        const uint64_t m = __builtin_amdgcn_ballot_w64(ok);
        if (ok) list[cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = t;
        cnt += (uint32_t)__builtin_popcountll(m);
      }
      if (cnt == 0) continue;
      visited += cnt;
      // Reference tile data in two register buffers (the loads run one survivor ahead); the chains
      // are software-pipelined over two accumulator tiles: while the MFMAs of one chain run, the
      // tile minimum of the previous chain issues in their shadow (a wave issues in order).
      s16x8 a0[NM];
      auto entry = [&](uint32_t i) {
        return (uint32_t)__builtin_amdgcn_readfirstlane(list[i < cnt ? i : cnt - 1]);
      };
      // the rest of an epilogue: free-energy classes, band test, parking of the candidates.
      // (t, fr) describe the reference tile the accumulator belongs to.
      auto finish = [&](const f32x16& acc, auto qi_c, float tmin, uint32_t t, float2 fr) __attribute__((always_inline)) {
        constexpr int qi = decltype(qi_c)::value;
        NnPQr& Q = q[qi];
        // Common path: two compares against the cached candidate thresholds.  "Lower free energy" is
        // taken conservatively here (the tile has SOME lower frame => its minimum might be one), and
        // the tile holding the query itself always passes (its own d2 ~ 0): whatever needs the
        // per-element treatment ends up in the rare path.  The running minima can only change there
        // too (a value below the minimum is below its band).  ONE wave-level test: the scalar
        // hand-off (v_cmp -> s_cbranch) is a pipeline bubble at two waves per SIMD.
        // (bh >= bn always -- the minimum over the lower-free-energy frames cannot undercut the minimum over all
        //  frames -- so a tile that has lower frames is tested against bh alone, any other against bn)
        const float thr = (fr.x < Q.feq) ? Q.bh : Q.bn;
        const bool rare = tmin < thr;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(rare) != 0, 0)) {
          const bool all_lower = fr.y < Q.feq;
          const bool mixed = (fr.x < Q.feq) & !all_lower;
          const bool special = mixed | (t == (Q.spos >> 5));
          float hmin = all_lower ? tmin : INFINITY;
          const bool any_special = __builtin_amdgcn_ballot_w64(special) != 0;
          if (any_special) {
            // masked per-element minima (the tile holds the query itself and/or straddles feq); the
            // free energies of the tile's frames are fetched only here
            float4 fv[4];
            load_frag(fe_c, t, h, fv);
            const f32x16 fef = frag16(fv);
            tmin = INFINITY;
            hmin = INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float v = (tile_row(t, r, h) != Q.spos) ? acc[r] : INFINITY;
              tmin = fminf(tmin, v);
              hmin = fminf(hmin, (fef[r] < Q.feq) ? v : INFINITY);
            }
          }
          // (the two half-wave lanes of a query see different rows of every tile: what either of them has
          //  found bounds the answer of both, so the running minima are shared whenever they move -- the
          //  records of one sequence over all rows instead of two over half of them each)
          float new_nn = fminf(Q.m_nn, nn_unprime(tmin, cq[qi])), new_hd = fminf(Q.m_hd, nn_unprime(hmin, cq[qi]));
          if constexpr (COOP) {
            // what the other shares of the group have learnt in the meantime, and what this chain adds to it (a returning
            // ds_min per kind; idle lanes carry -inf on both sides)
            const uint32_t o_nn = atomicMin(&smin[qi * 32 + c], fkey(new_nn));
            const uint32_t o_hd = atomicMin(&smin[TQ * 32 + qi * 32 + c], fkey(new_hd));
            new_nn = fminf(new_nn, fkey_inv(o_nn));
            new_hd = fminf(new_hd, fkey_inv(o_hd));
          }
          new_nn = fminf(new_nn, __shfl_xor(new_nn, 32, 64));
          new_hd = fminf(new_hd, __shfl_xor(new_hd, 32, 64));
          const float bn = nn_prime(nn_band(gb, new_nn), cq[qi]), bh = nn_prime(nn_band(gb, new_hd), cq[qi]);
          const bool trig = (tmin < bn) | (hmin < bh);
          if (__builtin_amdgcn_ballot_w64(trig) != 0) {
            // park this tile's candidates (values within the band of the running minima); element r
            // of the accumulator is bit (15 - r) of the masks
            uint32_t mn = 0, mh = 0;
            if (!any_special && t + 1 != T) {
              // plain tile: below-threshold sign strings (idle lanes have thresholds of -inf, pad
              // rows only exist in the last tile)
              uint32_t sn = 0, sh = 0;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                sn = __builtin_amdgcn_alignbit(sn, __float_as_uint(acc[r] - bn), 31);
                sh = __builtin_amdgcn_alignbit(sh, __float_as_uint(acc[r] - bh), 31);
              }
              mn = sn & 0xFFFFu;
              mh = all_lower ? (sh & 0xFFFFu) : 0u;
            } else {
              float4 fv[4];
              load_frag(fe_c, t, h, fv);
              const f32x16 fef = frag16(fv);
              const bool live = (livemask[qi] >> lane) & 1;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const uint32_t pos = tile_row(t, r, h);
                const bool other = live & (pos != Q.spos) & (pos < CV.n_pos);
                mn |= (other & (acc[r] < bn)) ? (0x8000u >> r) : 0u;
                mh |= (other & (acc[r] < bh) & (fef[r] < Q.feq)) ? (0x8000u >> r) : 0u;
              }
            }
            uint32_t m = mn | mh;
            for (;;) {
              const uint64_t have = __builtin_amdgcn_ballot_w64(m != 0);
              if (have == 0) break;
              const uint32_t n_new = (uint32_t)__builtin_popcountll(have);
              if (qn + n_new > (uint32_t)kWaveQueue) flush();
              if (m != 0) {
                const int p = __builtin_ctz(m);
                const uint32_t slot = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
                cand[slot] = make_uint2(tile_row(t, 15 - p, h) | (((mn >> p) & 1u) << 30) | (((mh >> p) & 1u) << 31),
                                        (uint32_t)(qi * 32 + c));
                m &= m - 1;
              }
              qn += n_new;
            }
            if (qn >= 64u) flush();
          }
          Q.m_nn = new_nn;
          Q.m_hd = new_hd;
          Q.bn = bn + skipb;
          Q.bh = bh + skipb;
        }
      };
      // Full chains (NM = 1 and the single-buffer instances): accB always holds the chain whose epilogue is still
      // pending -- query tile TQ-1 of reference tile tB (or +inf everywhere: no minimum, no candidates).  The early-out
      // form (kNnEarly) keeps nothing pending across tiles and uses accA / accB as its two coarse accumulators.
      f32x16 accA, accB;
#pragma unroll
      for (int r = 0; r < 16; ++r) accB[r] = INFINITY;
      uint32_t tB = 0;
      float2 frB = make_float2(INFINITY, INFINITY);
      // (fr: the tile's free-energy range, fetched with its operands one tile ahead -- read at the start of the
      //  tile's own chains the scalar load's latency sat in front of the second chain of every tile)
      auto compute = [&](s16x8 (&a)[NM], uint32_t t, uint32_t t_next, float2 fr) {
        f32x16 c0;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = 0.0f;   // (an inline constant of the first MFMA)
        chains += TQ;
        static_assert(TQ % 2 == 0, "accumulator ping-pong needs an even number of query tiles");
        if constexpr (kNnEarly<NM>) {
          // Early-out (see nn_chain_coarse): the coarse minima of the tile's TQ chains are tested TOGETHER -- one scalar
          // hand-off per reference tile instead of one per chain (C3: 11.1 -> 10.8 ms) --, and a chain that goes on is
          // computed again from its first MFMA, so no accumulator has to wait for its test and nothing stays pending
          // across tiles.
          float tm[TQ], dmin = INFINITY;
          auto coarse_first = [&](f32x16& acc, const s16x8 (&bq)[NM]) {
            acc = mfma16(a[0], bq[0], c0);
#pragma unroll
            for (int m = 1; m < kNnCoarse<NM>; ++m) acc = mfma16(a[m], bq[m], acc);
          };
          coarse_first(accA, b[0]);
          constexpr_for_pairs<TQ>([&](auto qt_c) {
            constexpr int qt = decltype(qt_c)::value;
            tm[qt] = INFINITY;
            nn_chain_coarse<NM, kNnCoarse<NM>, NM>(a, b[qt + 1], c0, accB, accA, tm[qt]);
            tm[qt + 1] = INFINITY;
            if constexpr (qt + 2 < TQ)
              nn_chain_coarse<NM, kNnCoarse<NM>, NM>(a, b[qt + 2], c0, accA, accB, tm[qt + 1]);
            else
              tile_min<0, 16>(accB, tm[qt + 1]);
          });
#pragma unroll
          for (int qi = 0; qi < TQ; ++qi) dmin = fminf(dmin, tm[qi] - ((fr.x < q[qi].feq) ? q[qi].bh : q[qi].bn));
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(dmin < 0.0f) != 0, 0)) {
            constexpr_for_all<TQ>([&](auto qi_c) {
              constexpr int qi = decltype(qi_c)::value;
              const float thr_c = (fr.x < q[qi].feq) ? q[qi].bh : q[qi].bn;
              if (__builtin_amdgcn_ballot_w64(tm[qi] < thr_c) != 0) {
                chains_on += 1;
                f32x16 acc = mfma16(a[0], b[qi][0], c0);
#pragma unroll
                for (int m = 1; m < NM; ++m) acc = mfma16(a[m], b[qi][m], acc);
                float tmin = INFINITY;
                tile_min<0, 16>(acc, tmin);
                finish(acc, qi_c, tmin, t, fr);
              }
            });
          }
          return;
        }
        auto refill = [&](auto mi_c) {
          if constexpr (kSingleBuffer<NM>) {
            constexpr int MI = decltype(mi_c)::value;
            const uint4 v = img_r[(size_t)t_next * (NM * 64) + MI * 64 + lane];
            a[MI] = __builtin_bit_cast(s16x8, v);
          }
        };
        constexpr_for_pairs<TQ>([&](auto qt_c) {
          constexpr int qt = decltype(qt_c)::value;
          constexpr int qb = (qt == 0) ? TQ - 1 : qt - 1;
          float tmin = INFINITY;
          nn_chain<NM>(a, b[qt], c0, accA, accB, tmin);
          finish(accB, std::integral_constant<int, qb>{}, tmin, (qt == 0) ? tB : t,
                 (qt == 0) ? frB : fr);
          tmin = INFINITY;
          if constexpr (qt + 2 == TQ)   // last chain of the tile
            nn_chain<NM>(a, b[qt + 1], c0, accB, accA, tmin, refill);
          else
            nn_chain<NM>(a, b[qt + 1], c0, accB, accA, tmin);
          finish(accA, std::integral_constant<int, qt>{}, tmin, t, fr);
        });
        tB = t;
        frB = fr;
      };
      if constexpr (kSingleBuffer<NM>) {
        uint32_t t0 = entry(0);
        load_tile_folded<NM>(img_r, t0, lane, a0);
        float2 f0 = ferange_r[t0];
        for (uint32_t i = 0; i < cnt; ++i) {
          const uint32_t t1 = entry(i + 1);
          const float2 f1 = ferange_r[t1];
          compute(a0, t0, t1, f0);
          t0 = t1;
          f0 = f1;
        }
      } else {
        s16x8 a1[NM];
        // (the survivor list is read one tile ahead of its use: the LDS latency sat in front of every tile's loads)
        auto peek = [&](uint32_t i) { return list[i < cnt ? i : cnt - 1]; };
        uint32_t t0 = entry(0), t1;
        uint32_t l_next = peek(1);
        load_tile_folded<NM>(img_r, t0, lane, a0);
        float2 f0 = ferange_r[t0], f1;
        for (uint32_t i = 0; i < cnt; i += 2) {
          t1 = (uint32_t)__builtin_amdgcn_readfirstlane(l_next);
          l_next = peek(i + 2);
          load_tile_folded<NM>(img_r, t1, lane, a1);
          f1 = ferange_r[t1];
          compute(a0, t0, t1, f0);
          if (i + 1 < cnt) {
            t0 = (uint32_t)__builtin_amdgcn_readfirstlane(l_next);
            l_next = peek(i + 3);
            load_tile_folded<NM>(img_r, t0, lane, a0);
            f0 = ferange_r[t0];
            compute(a1, t1, t0, f1);
          }
        }
      }
      {  // drain: epilogue of the last pending chain of this round
        float tmin = INFINITY;
        tile_min<0, 16>(accB, tmin);
        if constexpr (kNnEarly<NM>)
          (void)tmin;   // (nothing stays pending across tiles)
        else
          finish(accB, std::integral_constant<int, TQ - 1>{}, tmin, tB, frB);
      }
    }
    flush();                                          // the settle test needs the exact incumbents (in LDS)
    if (!(r2_hi <= FLT_MAX) || visited >= U - u_lo)
      break;   // every reference tile of this wave's share has been visited
    // settled: every unvisited frame is >= sqrt(r2_hi) away; the exact incumbents decide
    const float sure = r2_hi * 0.9999f;
    // The next ring must cover the WORST open query of the group: the largest incumbent still to be confirmed.
    float need = 0.0f;
#pragma unroll
    for (int qt = 0; qt < TQ; ++qt) {
      const bool live = (livemask[qt] >> lane) & 1;
      const bool hd_possible = fe_floor < q[qt].feq;
      // (both half-wave lanes of a query read the same words)
      const float inc_nn = fminf(g_pub[qt * 32 + c], __uint_as_float((uint32_t)(best64[qt * 32 + c] >> 32)));
      const float inc_hd = fminf(g_pub[TQ * 32 + qt * 32 + c], __uint_as_float((uint32_t)(best64[TQ * 32 + qt * 32 + c] >> 32)));
      const float want = fminf(fmaxf(inc_nn, hd_possible ? inc_hd : 0.0f), 3.0e38f);   // (no candidate at all: 3e38)
      const bool open = live & (h == 0) & !(want < sure);
      need = fmaxf(need, open ? want : 0.0f);
    }
    need = wave_max(need);
    if (!(need > 0.0f)) break;   // every query of the group is settled
    r2_lo = r2_hi;
    if (need >= 1.0e38f) {
      r2_hi = r2_hi * 4.0f;   // (a query without any candidate yet)
    } else {
      // (an intermediate ring at 0.35 .. 0.7 of this radius, to tighten the incumbents before the confirming
      //  ring, was measured 3 % slower at C3: the incumbents of the first ring are already close to final)
      r2_hi = fmaxf(need * 1.001f, r2_hi * 1.001f);
    }
    if (!(r2_hi < 1.0e37f)) r2_hi = INFINITY;
  }
  if (lane == 0 && chain_counter) {
    atomicAdd(chain_counter, (unsigned long long)chains);
    atomicAdd(chain_counter + kMfmaCtrNn, kNnEarly<NM> ? (unsigned long long)chains * kNnCoarse<NM> + (unsigned long long)chains_on * NM
                                                       : (unsigned long long)chains * NM);
  }

#pragma unroll
  for (int qt = 0; qt < TQ; ++qt) {
    if (h == 0 && ((livemask[qt] >> lane) & 1)) {
      const unsigned long long w_nn = best64[qt * 32 + c], w_hd = best64[TQ * 32 + qt * 32 + c];
      if (n_chunks == 1) {
        nn_idx[jq[qt]] = (uint32_t)w_nn;
        nn_d2[jq[qt]] = __uint_as_float((uint32_t)(w_nn >> 32));
        hd_idx[jq[qt]] = (uint32_t)w_hd;
        hd_d2[jq[qt]] = __uint_as_float((uint32_t)(w_hd >> 32));
      } else {
        // d2 >= 0, so (d2 bits << 32 | frame id) orders like the lexicographic (d2, id): the merge
        // over the chunks is a 64-bit atomic min (merge64 was filled with (FLT_MAX, n_rows+1))
        atomicMin(&merge64[jq[qt]], w_nn);
        atomicMin(&merge64[(size_t)n_rows + jq[qt]], w_hd);
      }
    }
  }
}

#include "dc_mfma_nn_shared.hpp"

__global__ void nn_merge_fill_kernel(unsigned long long* __restrict__ merge64, uint32_t n_rows) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * n_rows) merge64[i] = ((unsigned long long)__float_as_uint(FLT_MAX) << 32) | (n_rows + 1);
}

__global__ void nn_merge_unpack_kernel(const unsigned long long* __restrict__ merge64,
                                       const uint32_t* __restrict__ perm_q, uint32_t n_q,
                                       uint32_t n_rows, uint32_t tq, QSeg q_seg,
                                       uint32_t* __restrict__ nn_idx, float* __restrict__ nn_d2,
                                       uint32_t* __restrict__ hd_idx, float* __restrict__ hd_d2) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_q) return;
  // (rows of other segments keep what the caller initialised them with)
  if (!seg_owns(p / (32u * tq), q_seg)) return;
  const uint32_t i = perm_q[p];   // the query rows of this call
  if (i == kInvalidFrame) return; // (a pad position of the order)
  const unsigned long long a = merge64[i], b = merge64[(size_t)n_rows + i];
  nn_idx[i] = (uint32_t)a;
  nn_d2[i] = __uint_as_float((uint32_t)(a >> 32));
  hd_idx[i] = (uint32_t)b;
  hd_d2[i] = __uint_as_float((uint32_t)(b >> 32));
}

// the same by FRAME (coalesced) when the queries are the rows of the reference order: frame i sits at
// position invpos[i], which tells whether one of this launch's groups owns it
__global__ void nn_merge_unpack_rows_kernel(const unsigned long long* __restrict__ merge64,
                                            const uint32_t* __restrict__ invpos, uint32_t n_rows,
                                            uint32_t tq, QSeg q_seg, uint32_t* __restrict__ nn_idx,
                                            float* __restrict__ nn_d2, uint32_t* __restrict__ hd_idx,
                                            float* __restrict__ hd_d2) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  if (q_seg.stride > 1 && !seg_owns(invpos[i] / (32u * tq), q_seg)) return;
  const unsigned long long a = merge64[i], b = merge64[(size_t)n_rows + i];
  nn_idx[i] = (uint32_t)a;
  nn_d2[i] = __uint_as_float((uint32_t)(a >> 32));
  hd_idx[i] = (uint32_t)b;
  hd_d2[i] = __uint_as_float((uint32_t)(b >> 32));
}

// ---------------------------------------------------------------------------------------------
// launch helpers (one MFMA count per translation unit; the template parameter S below is NM)
// ---------------------------------------------------------------------------------------------
// query tiles per wave (pruned sweeps): as many as the resident B fragments (4 * TQ * NM registers)
// leave room for at two waves per SIMD; each reference fragment is fetched once per TQ chains.  Measured at
// 300k rows: NM = 5 (D = 24) 5.6 / 6.2 ms with four tiles against 6.4 / 6.5 with two; NM = 6 (D = 30) 7.8 / 9.3 against
// 7.1 / 7.4; NM = 7 (D = 32) 8.2 / 10.0 against 7.7 / 8.3; NM = 8 (D = 40) 11.0 / 11.4 against 8.8 / 10.0.
template <int NM>
constexpr int tq_for = (NM <= 5) ? 4 : 2;

// the population sweep keeps less state per query tile: with two MFMAs per chain six tiles fit
// (measured at C3: 23.6 ms against 25.1 ms with four; eight spill; the neighbour sweep loses at six)
template <int NM>
constexpr int tq_pop_for = (NM <= 2) ? 6 : tq_for<NM>;
// the neighbour sweeps: six as well since the early-out (round 4: the chains are short, the per-tile work -- loads,
// list, the one test -- is shared by six of them; C3 11.0 -> 10.7 ms, 256 registers, no spill.  DC_NN_TQ_SMALL=4:
// measurements)
#ifndef DC_NN_TQ_SMALL
#define DC_NN_TQ_SMALL 6
#endif
template <int NM>
constexpr int tq_nn_for = (NM <= 2) ? DC_NN_TQ_SMALL : tq_for<NM>;   // (dc_mfma.hip tq_nn_of)
// query tiles per wave of the full sweeps (double-buffered reference operands)
template <int NM>
constexpr int tq_full_for = (NM <= 4) ? 4 : (NM <= 8) ? 2 : 1;

inline uint32_t grid_for(uint32_t i_from, uint32_t i_to, int tq) {
  const uint32_t tiles = (i_to + 31) / 32 - i_from / 32;
  const uint32_t waves = (tiles + tq - 1) / tq;
  return (waves + 3) / 4;
}

template <int S>
void pop_dispatch(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P, uint32_t T,
                  uint32_t i_from, uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* pops,
                  hipStream_t s) {
  constexpr int kTQ = tq_full_for<S>;
  const dim3 grid(grid_for(i_from, i_to, kTQ)), block(256);
  // one radius per sweep (dc_mfma.hip loops over the radii of a call)
  { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_mfma_kernel<S, 1, kTQ>), grid, block, 0, s, coords, n_rows, n_cols, P.img,
                     P.img_b, P.norms, P.hdr, T, i_from, i_to, rad2, n_rad, pops); sweep_timer_mark(0, false, s); }
}

// tile boxes regrouped by reference share: share c holds the tiles c, c + n, c + 2n, ... at
// box_t[c * ceil(T / n) + u]
__global__ void box_by_share_kernel(const float4* __restrict__ box_r, uint32_t T, uint32_t n_chunks,
                                    float4* __restrict__ box_t) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const uint32_t stride = (T + n_chunks - 1) / n_chunks;
  box_t[(size_t)(t % n_chunks) * stride + t / n_chunks] = box_r[t];
}

struct NnPrunedArgs {     // regions of the neighbour sweep's (cell, free energy) ordering
  const uint4* img_r;
  const float* norms_r;
  const uint32_t* perm_r;
  const float4* box_r;
  float4* box_t;
  const float2* ferange_r;
  const float* fe_c;
  const float* coords_c;
  const uint32_t* invpos_r;
  const uint4* img_q;
  const float* norms_q;
  const uint32_t* perm_q;
  const float4* box_q;
  unsigned long long* merge64;
  uint32_t n_q;
  QSeg q_seg;
  int full_range;
  float cell2;
  const uint32_t* tile_comp_q;   // component of every tile of the query order
  const uint32_t* comp;          // component region
};

// Reference chunks per query group (gridDim.y).  The cost of a query group follows the local density
// of the data (dense regions keep many more reference tiles), and with two resident waves per SIMD a
// launch of a few thousand waves ends in a long, mostly idle tail behind its heaviest groups.  The
// launch is therefore split along the reference axis into about `target` waves; the operand reuse of
// TQ query tiles per wave is kept, partial results merge with atomics.  A share never drops below
// `share_floor` reference tiles: the per-wave set-up, the ring logic of the neighbour sweep and the
// box scans must stay small next to the chains.  Measured on C3 (1M x 10): the full sweeps are
// fastest at 12 chunks (pop 38.3 -> 35.6 ms, nn 50.2 -> 38.8 ms against one chunk), one eighth of the
// rows (one rank of an 8-GPU run) at 17..64 (pop) / 34 (nn) chunks.
constexpr uint32_t kPopWaveTarget = 49152, kPopSharedWaveTarget = 196608, kNnWaveTarget = 98304, kNnWaveTargetPerWave = 40960;   // (round 3, with the component-wise scans: pop 98304 / 512 ->
constexpr uint32_t kNnCoopMinShares = 16;   // shares per group from which the neighbour sweep's shares are waves of one workgroup
constexpr uint32_t kPopShareFloor = 1024, kNnShareFloor = 900;     //  49152 / 1024: C3 12.30 -> 12.12 ms, one eighth of it 1.82 -> 1.72 ms;
                                                                   //  per-wave neighbour sweep 98304 -> 57344: C3 14.26 -> 14.10 ms, two boxes;
                                                                   //  round 4, six query tiles per wave: 57344 -> 40960, 8 shares instead of 12
                                                                   //  at C3, 11.2 -> 11.0 ms on one box;
                                                                   //  round 4, shared-operand population sweeps 49152 -> 196608: they end with
                                                                   //  fewer half-empty CUs -- one rank of C5 450 -> 409 ms, 1M x 30 x 8 radii
                                                                   //  154 -> 132, 1M x 16 x 4 radii 91 -> 78, 600k x 40 19.2 -> 17.3 ms; the
                                                                   //  per-wave sweep of C3 is box-dependent at 196608 (11.76 -> 11.58 on one,
                                                                   //  12.1 -> 12.4 ms on another, 56 % more memory-side traffic): it stays)
// waves per workgroup of the per-wave sweeps (pop_pruned_kernel, nn_pruned_kernel; see nn_pruned_kernel): ONE while a
// chain has at most two MFMAs -- no slot waits for the slowest wave of a workgroup (1M x 10: neighbours 16.1 -> 15.0 ms,
// 1M x 3: 5.7 -> 5.3 / populations 6.3 -> 6.05 ms) -- and four beyond that: the four waves of a workgroup sit on one CU
// and walk nearly the same reference tiles at nearly the same time, so three of them find the operands in its L1
// (populations 600k x 12: 8.0 ms with four waves per workgroup, 10.4 ms with one; 300k x 40: 6.1 / 7.9 ms; the neighbour
// sweep does not care).  DC_WAVES_PER_GROUP = 1 / 2 / 4 for measurements.
inline uint32_t waves_per_group(int nm) {
  static const int forced = [] {
    const char* v = getenv("DC_WAVES_PER_GROUP");
    const int k = (v && v[0]) ? atoi(v) : 0;
    return (k == 1 || k == 2 || k == 4) ? k : 0;
  }();
  if (forced) return (uint32_t)forced;
  return nm <= 2 ? 1u : 4u;
}
// Share floor of the per-wave population sweep by problem size: a share of 1 024 reference tiles is right from about
// 400 000 rows on (C3), but it left a 100 000-row problem (C2: 3 500 tiles) with three shares = 1 500 waves for the
// chip's 2 048 wave slots -- 1.10 ms per step against 0.89 ms at 256 tiles per share (128: 1.02 ms, the per-wave
// set-up takes over).  A twelfth of the reference tiles, between 256 and 1 024.
inline uint32_t pop_share_floor(uint32_t ref_tiles) {
  const uint32_t f = ref_tiles / 12u;
  return f < 256u ? 256u : (f > kPopShareFloor ? kPopShareFloor : f);
}
inline uint32_t pick_chunks(uint32_t tiles, int tq, uint32_t target, uint32_t ref_tiles,
                            uint32_t share_floor, size_t /*tile_bytes*/) {
  // DC_WAVE_TARGET / DC_SHARE_FLOOR: measurement overrides of the two tuning constants (both sweeps)
  static const uint32_t env_target = [] { const char* v = getenv("DC_WAVE_TARGET"); return (v && v[0]) ? (uint32_t)atoi(v) : 0u; }();
  static const uint32_t env_floor = [] { const char* v = getenv("DC_SHARE_FLOOR"); return (v && v[0]) ? (uint32_t)atoi(v) : 0u; }();
  if (env_target) target = env_target;
  if (env_floor) share_floor = env_floor;
  const uint32_t waves = (tiles + tq - 1) / tq;
  uint32_t r = waves >= target ? 1u : (target + waves - 1) / waves;
  const uint32_t by_share = ref_tiles / share_floor;
  const uint32_t cap = by_share < 64u ? by_share : 64u;
  r = r > cap ? cap : r;
  return r < 1u ? 1u : r;
}

template <int S, int TQV>
void nn_pruned_launch(const float* coords, uint32_t n_rows, uint32_t n_cols, const float* fe,
                      const NnPrunedArgs& A, uint32_t T, const uint32_t* hdr,
                      unsigned long long* chain_counter, uint32_t* nn_idx, float* nn_d2,
                      uint32_t* hd_idx, float* hd_d2, hipStream_t s) {
  if (A.n_q == 0) return;
  // T: tiles of the (padded) reference order; A.n_q: positions of the query order
  const CompView CV{A.tile_comp_q, A.comp + kCompRange, 32u * T, A.comp};
  if (nn_shared_wanted(n_rows, n_cols)) {
    // reference operands shared through LDS (dc_mfma_nn_shared.hpp): the workgroup's 4 * TQV tiles are one group
    const uint32_t groups = seg_groups(((A.n_q + 31) / 32 + 4 * TQV - 1) / (4 * TQV), A.q_seg);
    if (groups == 0) return;
    const uint32_t n_chunks = pick_chunks(groups * 4 * TQV, TQV, kNnWaveTarget, T, kNnShareFloor, (size_t)S * 1024 + 128);
    const size_t smem = (size_t)kRing * (S * 64 + kNnRingExtra) * 16 + sizeof(uint32_t) * 4 * (TQV * kQueueCap * 64 + TQV * 32);
    if (n_chunks > 1)
      hipLaunchKernelGGL(nn_merge_fill_kernel, dim3((2 * n_rows + 255) / 256), dim3(256), 0, s, A.merge64, n_rows);
    hipLaunchKernelGGL(box_by_share_kernel, dim3((T + 255) / 256), dim3(256), 0, s, A.box_r, T, n_chunks, A.box_t);
    {
      sweep_timer_mark(1, true, s);
      // (the MFMAs in front of the early-out test: the most this NM can need, or one fewer for its narrowest rows)
      constexpr int kNbMax = kNnCoarseMax<S>;
      constexpr int kNbMin = (S > 1) ? nn_coarse_for((16 * (S - 1) - kConstSlots) / kPieceGroups + 1) : kNbMax;
      auto launch = [&](auto nb_c) {
        hipLaunchKernelGGL((nn_shared_kernel<S, TQV, decltype(nb_c)::value>), dim3(grid_x8(groups), n_chunks), dim3(256), smem, s, coords,
                           n_rows, n_cols, fe, A.img_r, A.norms_r, A.perm_r, A.box_r, (const float4*)A.box_t, A.ferange_r, A.fe_c,
                           A.coords_c, A.invpos_r, T, A.img_q, A.norms_q, A.perm_q, A.box_q, A.n_q, A.q_seg, A.full_range, A.cell2,
                           hdr, chain_counter, A.merge64, nn_idx, nn_d2, hd_idx, hd_d2, CV);
      };
      if constexpr (kNbMin != kNbMax) {
        if (nn_coarse_for((int)n_cols) < kNbMax)
          launch(std::integral_constant<int, kNbMin>{});
        else
          launch(std::integral_constant<int, kNbMax>{});
      } else {
        launch(std::integral_constant<int, kNbMax>{});
      }
      sweep_timer_mark(1, false, s);
    }
    if (n_chunks > 1 && A.full_range)
      hipLaunchKernelGGL(nn_merge_unpack_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s,
                         (const unsigned long long*)A.merge64, A.invpos_r, n_rows, (uint32_t)(4 * TQV), A.q_seg, nn_idx,
                         nn_d2, hd_idx, hd_d2);
    else if (n_chunks > 1)
      hipLaunchKernelGGL(nn_merge_unpack_kernel, dim3((A.n_q + 255) / 256), dim3(256), 0, s,
                         (const unsigned long long*)A.merge64, A.perm_q, A.n_q, n_rows, (uint32_t)(4 * TQV), A.q_seg, nn_idx,
                         nn_d2, hd_idx, hd_d2);
    return;
  }
  const uint32_t waves = seg_groups(((A.n_q + 31) / 32 + TQV - 1) / TQV, A.q_seg), tiles = waves * TQV;
  if (waves == 0) return;
  uint32_t n_chunks = pick_chunks(tiles, TQV, kNnWaveTargetPerWave, T, kNnShareFloor, (size_t)S * 1024 + 128);
  // Many shares per group (a rank of a sharded run: 34 at an eighth of C3): the shares of a group as the waves of ONE
  // workgroup that learn their thresholds together (nn_pruned_kernel<.., COOP>); DC_NN_COOP = 0 / 1 forces either form
  static const int coop_env = [] { const char* v = getenv("DC_NN_COOP"); return (v && v[0]) ? atoi(v) : -1; }();
  static const uint32_t kCoopWaves = [] { const char* v = getenv("DC_NN_COOP_WAVES"); const int k = (v && v[0]) ? atoi(v) : 4; return (k == 2 || k == 4 || k == 8) ? (uint32_t)k : 4u; }();
  const bool coop = (coop_env >= 0) ? (coop_env != 0 && n_chunks >= kCoopWaves) : (n_chunks >= kNnCoopMinShares);
  if (coop) {
    n_chunks = (n_chunks + kCoopWaves - 1) / kCoopWaves * kCoopWaves;
    const size_t smem_c = sizeof(uint32_t) * (kCoopWaves * (kListCap + 2 * kWaveQueue) + 8 * TQV * 32) +
                          sizeof(float) * TQV * 32 * (size_t)n_cols;
    hipLaunchKernelGGL(nn_merge_fill_kernel, dim3((2 * n_rows + 255) / 256), dim3(256), 0, s, A.merge64, n_rows);
    hipLaunchKernelGGL(box_by_share_kernel, dim3((T + 255) / 256), dim3(256), 0, s, A.box_r, T, n_chunks, A.box_t);
    sweep_timer_mark(1, true, s);
    hipLaunchKernelGGL((nn_pruned_kernel<S, TQV, true>), dim3(grid_x8(waves), n_chunks / kCoopWaves), dim3(64 * kCoopWaves), smem_c, s,
                       coords, n_rows, n_cols, fe, A.img_r, A.norms_r, A.perm_r, A.box_r, (const float4*)A.box_t, A.ferange_r,
                       A.fe_c, A.coords_c, A.invpos_r, T, A.img_q, A.norms_q, A.perm_q, A.box_q, A.n_q, A.q_seg,
                       A.full_range, A.cell2, hdr, chain_counter, A.merge64, nn_idx, nn_d2, hd_idx, hd_d2, CV);
    sweep_timer_mark(1, false, s);
    if (A.full_range)
      hipLaunchKernelGGL(nn_merge_unpack_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s,
                         (const unsigned long long*)A.merge64, A.invpos_r, n_rows, (uint32_t)TQV, A.q_seg,
                         nn_idx, nn_d2, hd_idx, hd_d2);
    else
      hipLaunchKernelGGL(nn_merge_unpack_kernel, dim3((A.n_q + 255) / 256), dim3(256), 0, s,
                         (const unsigned long long*)A.merge64, A.perm_q, A.n_q, n_rows, (uint32_t)TQV, A.q_seg,
                         nn_idx, nn_d2, hd_idx, hd_d2);
    return;
  }
  // query rows (original coordinates) + candidate queues, per wave
  const uint32_t wpb = waves_per_group(S);
  const size_t smem = wpb * (sizeof(uint32_t) * kListCap + sizeof(float) * TQV * 32 * (size_t)n_cols +
                             sizeof(uint32_t) * (TQV * kQueueCap * 64 + 2 * TQV * 32));
  if (n_chunks > 1)
    hipLaunchKernelGGL(nn_merge_fill_kernel, dim3((2 * n_rows + 255) / 256), dim3(256), 0, s,
                       A.merge64, n_rows);
  hipLaunchKernelGGL(box_by_share_kernel, dim3((T + 255) / 256), dim3(256), 0, s, A.box_r, T, n_chunks, A.box_t);
  { sweep_timer_mark(1, true, s); hipLaunchKernelGGL((nn_pruned_kernel<S, TQV>), dim3(grid_x8((waves + wpb - 1) / wpb), n_chunks), dim3(64 * wpb), smem, s,
                     coords, n_rows, n_cols, fe, A.img_r, A.norms_r, A.perm_r, A.box_r, (const float4*)A.box_t, A.ferange_r,
                     A.fe_c, A.coords_c, A.invpos_r, T, A.img_q, A.norms_q, A.perm_q, A.box_q, A.n_q, A.q_seg,
                     A.full_range, A.cell2, hdr, chain_counter, A.merge64, nn_idx, nn_d2, hd_idx,
                     hd_d2, CV); sweep_timer_mark(1, false, s); }
  if (n_chunks > 1 && A.full_range)
    hipLaunchKernelGGL(nn_merge_unpack_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s,
                       (const unsigned long long*)A.merge64, A.invpos_r, n_rows, (uint32_t)TQV, A.q_seg,
                       nn_idx, nn_d2, hd_idx, hd_d2);
  else if (n_chunks > 1)
    hipLaunchKernelGGL(nn_merge_unpack_kernel, dim3((A.n_q + 255) / 256), dim3(256), 0, s,
                       (const unsigned long long*)A.merge64, A.perm_q, A.n_q, n_rows, (uint32_t)TQV, A.q_seg,
                       nn_idx, nn_d2, hd_idx, hd_d2);
}

template <int S>
void nn_pruned_dispatch(const float* coords, uint32_t n_rows, uint32_t n_cols, const float* fe,
                        const NnPrunedArgs& A, uint32_t T, const uint32_t* hdr,
                        unsigned long long* chain_counter, uint32_t* nn_idx, float* nn_d2,
                        uint32_t* hd_idx, float* hd_d2, hipStream_t s) {
  // (the shared-operand sweep keeps tq_for: its workgroup of four waves is one group of 4 * TQ tiles)
  if (nn_shared_wanted(n_rows, n_cols))
    nn_pruned_launch<S, tq_for<S>>(coords, n_rows, n_cols, fe, A, T, hdr, chain_counter, nn_idx, nn_d2, hd_idx, hd_d2, s);
  else
    nn_pruned_launch<S, tq_nn_for<S>>(coords, n_rows, n_cols, fe, A, T, hdr, chain_counter, nn_idx, nn_d2, hd_idx, hd_d2, s);
}

// pruned population sweep: queries = n_q spatially ordered rows (image/perm/boxes "q"), references =
// all rows spatially ordered ("p"); full_range: the query set is every row -> the two orders coincide
// Which calls take the symmetric sweep: all rows as queries in the reference order -- every query group, or the
// groups of one segment of a sharded run (a group owns the same pairs of groups whichever rank runs it, so the
// ranks' counts are PARTIAL counts of all rows and merge by summation like the one-sided ones) -- but not a row
// range; one radius, no pair sink, positions that fit the queue entries.  DC_POP_SYM = 0 turns it off (tests,
// measurements).
inline bool pop_sym_wanted(bool sink, int q_mode, QSeg q_seg, uint32_t n_rows, int n_rad) {
  static const bool off = [] {
    const char* v = getenv("DC_POP_SYM");
    return v && v[0] == '0';
  }();
  (void)q_seg;
  return !off && !sink && q_mode == kQueryAll && n_rad == 1 && n_rows + 32u * kPadTiles <= kPopQueueMaxRows;
}
// the shared-operand sweeps in their symmetric form: with ONE radius (5M x 30, all rows: 1 482 -> 932 ms; 1M x 30:
// 56.3 -> 35.8 ms).  With several radii per sweep the reference side costs one 128-byte atomic per reference tile,
// wave and RADIUS into count arrays far larger than the L2 (C5: 8 x 20 MB): four radii at 1M x 16 still gain 9 %, but
// the eight radii of C5's segment sweep went from 797 to 2 062 ms -- about 10^9 such atomics per second is what the
// memory side takes.  DC_POP_SHARED_SYM = 0 / 1 / 2: never / one radius only (default) / always.
inline bool pop_shared_sym_wanted(int nr) {
  static const int mode = [] {
    const char* v = getenv("DC_POP_SHARED_SYM");
    return (v && v[0]) ? atoi(v) : 1;
  }();
  return mode >= 2 || (mode == 1 && nr == 1);
}
// counts by position in the sweep's order -> populations by frame (a flagged data set: the direct kernel writes)
__global__ void pops_by_frame_kernel(const uint32_t* __restrict__ pops_pos, const uint32_t* __restrict__ perm,
                                     uint32_t n_rows, const uint32_t* __restrict__ hdr, uint32_t* __restrict__ pops) {
  if (hdr[1] != 0) return;
  const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;   // n_rows here: positions of the (padded) order
  if (pos < n_rows && perm[pos] != kInvalidFrame) pops[perm[pos]] = pops_pos[pos];
}

template <int S, int NRV, int TQV>
void pop_pruned_launch(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P,
                       uint32_t T, uint32_t n_q, int q_mode, QSeg q_seg, const Rad2& rad2,
                       int n_rad, uint32_t* pops, unsigned long long* chain_counter,
                       const EdgeSink* sink, hipStream_t s, bool pos_clean = false) {
  // pos_clean: the counts by position of the one-radius symmetric per-wave sweep (the pq region) were cleared by the
  // preparation (order_rows2_kernel) -- no fill in front of the sweep
  if (n_q == 0) return;
  // the query groups of this launch: all of them, or one segment's share
  const uint32_t waves = seg_groups(((n_q + 31) / 32 + TQV - 1) / TQV, q_seg), tiles = waves * TQV;
  if (waves == 0) return;
  const uint32_t wpb = waves_per_group(S);
  const dim3 grid(grid_x8((waves + wpb - 1) / wpb), pick_chunks(tiles, TQV, kPopWaveTarget, T, pop_share_floor(T), (size_t)S * 1024 + 128)),
      block(64 * wpb);
  // B form of the query rows: its own image for a row range, else the B form of the rows in the
  // reference order (img_q)
  const bool own = q_mode == kQueryOwnOrder;
  // T: tiles of the (padded) reference order; n_q: positions of the query order
  const CompView CV{own ? P.tile_comp_q : P.tile_comp, P.comp + kCompRange, 32u * T, P.comp};
  // pairs between adjacent components (exact, after the matrix-core sweep): into the same counts
  auto cross = [&](uint32_t group_tiles, uint32_t* out, size_t stride, int by_position) {
    if (sink) return;   // (the sweeps that list pairs use components no pair can cross)
    launch_pop_cross(coords, n_cols, P.coords_p, P.perm_p, P.box_p, own ? P.perm_q : P.perm_p, own ? P.box_q : P.box_p,
                     CV.tile_comp_q, P.comp, (n_q + 31) / 32, group_tiles, q_seg, own ? 0 : 1, rad2, n_rad, P.hdr, out,
                     stride, by_position, s);
  };
  const uint4* img_q = P.img_q;
  const float* norms_q = own ? P.norms_q : P.norms_p;
  const uint32_t* perm_q = own ? P.perm_q : P.perm_p;
  const float4* box_q = own ? P.box_q : P.box_p;
  // survivor list + query rows (original coordinates) + queues of deferred exact evaluations, per wave
  const size_t smem = wpb * (sizeof(uint32_t) * kListCap + sizeof(float) * TQV * 32 * (size_t)n_cols +
                             sizeof(uint32_t) * TQV * kQueueCap * 64);
  if (!sink && pop_shared_wanted(n_rows, n_cols, n_rad)) {
    // reference operands shared through LDS (dc_mfma_shared.hpp); NRV radii in this one sweep
    constexpr int kTQS = tq_shared_for<S, NRV>;
    const uint32_t groups = seg_groups(((n_q + 31) / 32 + 4 * kTQS - 1) / (4 * kTQS), q_seg);
    if (groups == 0) return;
    const dim3 grid_s(grid_x8(groups), pick_chunks(groups * 4 * kTQS, kTQS, kPopSharedWaveTarget, T, kPopShareFloor, (size_t)S * 1024 + 128));
    const size_t smem_s = (size_t)kRing * kTileUnits<S> * 16 + sizeof(uint32_t) * 4 * shared_wave_words(kTQS, NRV);
    if constexpr (NRV > 1 && kTQS == kMsTQ) {
      if (pop_sym_wanted(false, q_mode, q_seg, n_rows, 1) && pop_msym_wanted()) {
        // symmetric form for several radii (dc_mfma_msym.hpp): counts by position [tile][NRV][32] in the regions from
        // norms_s to vals_in of the workspace (the population sweeps leave them alone once the orders are built)
        uint32_t* pops_pos = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(P.norms_s));
        (void)hipMemsetAsync(pops_pos, 0, sizeof(uint32_t) * 32 * (size_t)T * NRV, s);
        // (two launches: the instance with the thresholds taken off the accumulator in place, and the one that subtracts on
        //  the vector unit -- each looks at the scale and the radii and the one they do not ask for returns at once)
        sweep_timer_mark(0, true, s);
        hipLaunchKernelGGL((pop_msym_kernel<S, NRV, true>), grid_s, dim3(256), (msym_smem<S, NRV>()), s, coords, n_rows, n_cols,
                           P.img_p, P.norms_p, P.box_p, P.coords_p, T, img_q, norms_q, perm_q, box_q, n_q, q_seg, P.hdr,
                           chain_counter, rad2, n_rad, CV, pops_pos);
        hipLaunchKernelGGL((pop_msym_kernel<S, NRV, false>), grid_s, dim3(256), (msym_smem<S, NRV>()), s, coords, n_rows, n_cols,
                           P.img_p, P.norms_p, P.box_p, P.coords_p, T, img_q, norms_q, perm_q, box_q, n_q, q_seg, P.hdr,
                           chain_counter, rad2, n_rad, CV, pops_pos);
        sweep_timer_mark(0, false, s);
        cross(4u * kTQS, pops_pos, (size_t)NRV, 2);
        hipLaunchKernelGGL((pops_by_frame_ms_kernel<NRV>), dim3((32 * T + 255) / 256), dim3(256), 0, s, (const uint32_t*)pops_pos,
                           P.perm_p, 32u * T, n_rows, n_rad, P.hdr, pops);
        return;
      }
    }
    if (pop_sym_wanted(false, q_mode, q_seg, n_rows, 1) && pop_shared_sym_wanted(NRV)) {
      // symmetric form: counts by position, one array of 32 T words per radius (the regions from norms_s to
      // vals_in of the workspace: the population sweeps leave them alone once the orders are built)
      uint32_t* pops_pos = const_cast<uint32_t*>(reinterpret_cast<const uint32_t*>(P.norms_s));
      (void)hipMemsetAsync(pops_pos, 0, sizeof(uint32_t) * 32 * (size_t)T * NRV, s);
      { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_shared_kernel<S, kTQS, NRV, true>), grid_s, dim3(256), smem_s, s, coords, n_rows, n_cols,
                         P.img_p, P.norms_p, P.box_p, P.coords_p, T, img_q, norms_q, perm_q, box_q, n_q, q_seg, P.hdr,
                         chain_counter, rad2, n_rad, pops, own ? 0 : 1, CV, pops_pos); sweep_timer_mark(0, false, s); }
      cross(4u * kTQS, pops_pos, (size_t)32 * T, 1);
      for (int rr = 0; rr < n_rad; ++rr)
        hipLaunchKernelGGL(pops_by_frame_kernel, dim3((32 * T + 255) / 256), dim3(256), 0, s,
                           (const uint32_t*)(pops_pos + (size_t)rr * 32 * T), P.perm_p, 32u * T, P.hdr,
                           pops + (size_t)rr * n_rows);
      return;
    }
    { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_shared_kernel<S, kTQS, NRV>), grid_s, dim3(256), smem_s, s, coords, n_rows, n_cols, P.img_p,
                       P.norms_p, P.box_p, P.coords_p, T, img_q, norms_q, perm_q, box_q, n_q, q_seg, P.hdr,
                       chain_counter, rad2, n_rad, pops, own ? 0 : 1, CV); sweep_timer_mark(0, false, s); }
    cross(4u * kTQS, pops, (size_t)n_rows, 0);
    return;
  }
  if constexpr (NRV == 1 && TQV <= 6) {
    if (pop_sym_wanted(sink != nullptr, q_mode, q_seg, n_rows, n_rad)) {
      // symmetric sweep: counts by position (the pq region of the workspace: only the full neighbour sweep uses
      // it), then to the frames
      uint32_t* pops_pos = const_cast<uint32_t*>(P.pq);
      if (!pos_clean) (void)hipMemsetAsync(pops_pos, 0, sizeof(uint32_t) * 32 * (size_t)T, s);
      { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_pruned_kernel<S, 1, TQV, kSinkNone, true>), grid, block, smem, s, coords, n_rows,
                         n_cols, P.img_p, P.norms_p, P.perm_p, P.box_p, P.coords_p, T, img_q, norms_q,
                         perm_q, box_q, n_q, q_seg, P.hdr, chain_counter, rad2, n_rad, pops,
                         EdgeSink{nullptr, nullptr, 0, nullptr, nullptr, nullptr}, CV, pops_pos); sweep_timer_mark(0, false, s); }
      cross((uint32_t)TQV, pops_pos, (size_t)32 * T, 1);
      hipLaunchKernelGGL(pops_by_frame_kernel, dim3((32 * T + 255) / 256), dim3(256), 0, s, (const uint32_t*)pops_pos,
                         P.perm_p, 32u * T, P.hdr, pops);
      return;
    }
  }
  // radius-graph variants: all rows only (query positions = reference positions)
  if (sink && sink->best)
    { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_pruned_kernel<S, NRV, TQV, kSinkMinEdge>), grid, block, smem, s, coords,
                       n_rows, n_cols, P.img_p, P.norms_p, P.perm_p, P.box_p, P.coords_p, T, img_q,
                       norms_q, perm_q, box_q, n_q, q_seg, P.hdr, chain_counter, rad2, n_rad, pops, *sink, CV); sweep_timer_mark(0, false, s); }
  else if (sink)
    { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_pruned_kernel<S, NRV, TQV, kSinkPairs>), grid, block, smem, s, coords, n_rows,
                       n_cols, P.img_p, P.norms_p, P.perm_p, P.box_p, P.coords_p, T, img_q, norms_q,
                       perm_q, box_q, n_q, q_seg, P.hdr, chain_counter, rad2, n_rad, pops, *sink, CV); sweep_timer_mark(0, false, s); }
  else
    { sweep_timer_mark(0, true, s); hipLaunchKernelGGL((pop_pruned_kernel<S, NRV, TQV, kSinkNone>), grid, block, smem, s, coords, n_rows,
                       n_cols, P.img_p, P.norms_p, P.perm_p, P.box_p, P.coords_p, T, img_q, norms_q,
                       perm_q, box_q, n_q, q_seg, P.hdr, chain_counter, rad2, n_rad, pops,
                       EdgeSink{nullptr, nullptr, 0, nullptr, nullptr, nullptr}, CV); sweep_timer_mark(0, false, s); }
  cross((uint32_t)TQV, pops, (size_t)n_rows, 0);
}

template <int S, int NRV>
void pop_pruned_tq(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P, uint32_t T,
                   uint32_t n_q, int q_mode, QSeg q_seg, const Rad2& rad2, int n_rad,
                   uint32_t* pops, unsigned long long* chain_counter, const EdgeSink* sink,
                   hipStream_t s, bool pos_clean = false) {
  pop_pruned_launch<S, NRV, tq_pop_for<S>>(coords, n_rows, n_cols, P, T, n_q, q_mode, q_seg, rad2,
                                       n_rad, pops, chain_counter, sink, s, pos_clean);
}

template <int S>
void pop_pruned_dispatch(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P,
                         uint32_t T, uint32_t n_q, int q_mode, QSeg q_seg, const Rad2& rad2,
                         int n_rad, uint32_t* pops, unsigned long long* chain_counter,
                         const EdgeSink* sink, hipStream_t s, bool pos_clean = false) {
  // one radius per sweep (dc_mfma.hip loops over the radii of a call) -- except the shared-operand sweep of wide
  // rows, which takes up to eight (dc_mfma_shared.hpp; S >= 3 only: no instances for the narrow shapes)
  if constexpr (S >= 3 && S <= 8) {
    if (!sink && n_rad > 1 && pop_shared_wanted(n_rows, n_cols, n_rad)) {
      if (n_rad <= 4)
        pop_pruned_launch<S, 4, 2>(coords, n_rows, n_cols, P, T, n_q, q_mode, q_seg, rad2, n_rad, pops, chain_counter, sink, s);
      else
        pop_pruned_launch<S, 8, 2>(coords, n_rows, n_cols, P, T, n_q, q_mode, q_seg, rad2, n_rad, pops, chain_counter, sink, s);
      return;
    }
  }
  pop_pruned_tq<S, 1>(coords, n_rows, n_cols, P, T, n_q, q_mode, q_seg, rad2, n_rad, pops,
                      chain_counter, sink, s, pos_clean);
}

template <int S>
void nn_dispatch(const float* coords, uint32_t n_rows, uint32_t n_cols, const Ptrs& P, uint32_t T,
                 uint32_t i_from, uint32_t i_to, uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx,
                 float* hd_d2, hipStream_t s) {
  constexpr int kTQnn = tq_full_for<S>;
  const dim3 grid(grid_for(i_from, i_to, kTQnn)), block(256);
  { sweep_timer_mark(1, true, s); hipLaunchKernelGGL((nn_mfma_kernel<S, kTQnn>), grid, block, 0, s, coords, n_rows, n_cols, P.img_b,
                     P.norms, P.img_s, P.norms_s, P.perm, P.invpos, P.pq, P.hdr, T, i_from, i_to, nn_idx,
                     nn_d2, hd_idx, hd_d2); sweep_timer_mark(1, false, s); }
}

}  // namespace

// one translation unit per MFMA count NM = nm_for(n_cols) (dc_mfma_step.hip, -DDC_STEP=n) so the
// instances build in parallel; dc_mfma.hip switches over them.
#define DC_DECLARE_STEP(SV)                                                                      \
  void pop_mfma_step_##SV(const float* coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,     \
                          uint32_t i_from, uint32_t i_to, const Rad2& rad2, int n_rad,           \
                          uint32_t* pops, hipStream_t s);                                        \
  void pop_pruned_step_##SV(const float* coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,   \
                            uint32_t T_ref, uint32_t n_q, int q_mode, QSeg q_seg, const Rad2& rad2, \
                            int n_rad, uint32_t* pops, const EdgeSink* sink, hipStream_t s, bool pos_clean); \
  void nn_pruned_step_##SV(const float* coords, uint32_t n_rows, uint32_t n_cols, const float* fe, \
                           void* d_ws, uint32_t T_ref, uint32_t n_q, int q_mode, QSeg q_seg, float cell2, \
                           uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx, float* hd_d2,             \
                           hipStream_t s);                                                       \
  void nn_mfma_step_##SV(const float* coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,      \
                         uint32_t i_from, uint32_t i_to, uint32_t* nn_idx, float* nn_d2,         \
                         uint32_t* hd_idx, float* hd_d2, hipStream_t s);

}  // namespace dc
