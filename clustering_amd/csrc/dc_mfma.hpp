// dc_mfma.hpp -- matrix-core variants (fp16x2 Gram form on the f16 MFMA as a classifier + guard band +
// canonical fp32 re-check of the undecided pairs): host-side entry points.
#pragma once
#include "dc_common.hpp"

namespace dc {

// Optional second product of the pruned population sweep: the list of all unordered frame pairs
// within the radius (the radius graph that the reference's screening walks frame by frame,
// density_clustering.cpp:292-332).  Pairs are emitted as positions in the sweep's spatial order,
// (query position, reference position) with reference < query, i.e. every pair exactly once.
struct EdgeSink {
  uint2* edges;                    // nullptr: counting only
  unsigned long long* count;       // total number of pairs found (may exceed capacity)
  unsigned long long capacity;
  // "lightest outgoing pair" variant (one round of a Boruvka minimum spanning forest of the radius
  // graph): comp / rank per position of the sweep's order, best per component id (a position):
  // min over pairs {a in the component, b outside, d2 < r2} of (max(rank_a, rank_b) << 32 | min(..))
  const uint32_t* comp;
  const uint32_t* rank;
  unsigned long long* best;
};
enum SinkMode { kSinkNone = 0, kSinkPairs = 1, kSinkMinEdge = 2 };

// true if the MFMA kernels handle this n_cols
bool mfma_supports(size_t n_cols);
// bytes of device scratch (operand images, norms) for a problem size; 0 if unsupported
size_t mfma_workspace_bytes(size_t n_rows, size_t n_cols);
// builds the operand images of d_coords in the workspace; returns 0 on success
// pruned: a pruned sweep follows (one statistics pass; the component region of the workspace is zero-filled with the
// header unless stats_valid, in which case the statistics of an earlier sweep stay and the sweep clears the region itself)
int mfma_prepare(const float* d_coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,
                 bool natural_image,
                 hipStream_t stream, bool stats_valid = false, bool pruned = false, const float* d_fe = nullptr);
// (d_fe: the free energies of a pruned NEIGHBOUR call that claims DC_FLAG_STATS_VALID -- their range and the component
//  guard of that sweep are part of the claim's two launches; launch_nn_pruned[_segment](reuse_components) relies on it)
void launch_pop_mfma(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                     uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops_first_row,
                     void* d_ws, hipStream_t stream);
// DC_VARIANT_MFMA32 (dc_mfma32.hpp): the fp32-input MFMA instance, n_cols 9..10, every pair (needs mfma_prepare first)
bool mfma32_supports(size_t n_cols);
void launch_pop_mfma32(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from, uint32_t i_to,
                       const Rad2& rad2, int n_rad, uint32_t* d_pops_first_row, void* d_ws, hipStream_t stream);
void launch_nn_mfma32(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe, uint32_t i_from,
                      uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws,
                      hipStream_t stream);
// population sweep over spatially ordered frames with tile-pair pruning (needs mfma_prepare first)
// comp_clean: mfma_prepare(pruned) of this call has zero-filled the component region of the workspace
void launch_pop_pruned(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                       uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops_first_row,
                       void* d_ws, hipStream_t stream, bool comp_clean = false);
// pruned population sweep at squared radius r2 that also lists every unordered frame pair with
// canonical d2 < r2 (frame ids; d_pairs may be nullptr to count only); needs mfma_prepare first.
// *d_count: number of pairs found (> capacity: buffer too small), ~0 if the data was flagged.
void launch_radius_pairs(const float* d_coords, uint32_t n_rows, uint32_t n_cols, float r2,
                         uint32_t* d_pops, uint2* d_pairs, unsigned long long capacity,
                         unsigned long long* d_count, void* d_ws, hipStream_t stream);
// one segment of a sharded sweep: the query groups (TQ consecutive tiles of the sweep's order) are dealt out to the
// segments in BLOCKS of `block` consecutive groups, block-cyclically: segment `offset` of `stride` owns the groups g
// with (g / block) % stride == offset ({1, 0, 1}: all of them).  Dealing cyclically gives every segment the same mix of
// dense and sparse regions (contiguous eighths of C3 differed by up to 16 % in work); dealing whole blocks keeps the
// groups that run side by side on an XCD next to each other in the order, so that they stream the same reference tiles
// through its L2 (group by group -- block 1 -- an eighth of C3 read 3 x the bytes of the unsharded sweep per tile pair
// at the L2's memory side, L2 hit rate 0.76 against 0.91, and its kernels took 12 - 17 % more than an eighth).
struct QSeg {
  uint32_t stride, offset, block;
};
__host__ __device__ inline uint32_t seg_groups(uint32_t n_groups, QSeg q) {   // groups the segment owns
  const uint32_t cyc = q.block * q.stride, full = n_groups / cyc, rem = n_groups - full * cyc;
  const uint32_t lo = q.offset * q.block;
  return full * q.block + (rem > lo ? (rem - lo < q.block ? rem - lo : q.block) : 0u);
}
__host__ __device__ inline uint32_t seg_group(uint32_t unit, QSeg q) {        // the unit-th group of the segment
  return (unit / q.block) * (q.block * q.stride) + q.offset * q.block + unit % q.block;
}
__host__ __device__ inline bool seg_owns(uint32_t group, QSeg q) { return (group / q.block) % q.stride == q.offset; }
__host__ __device__ inline uint32_t seg_unit(uint32_t group, QSeg q) {        // (inverse of seg_group for an owned group)
  return (group / (q.block * q.stride)) * q.block + group % q.block;
}
// groups per block for n_segments segments (a constant: every rank derives the same deal; blocks of 4 .. 256 groups
// were measured at C3 and change nothing -- DESIGN.md 6)
constexpr uint32_t kSegBlockGroups = 1;
uint32_t seg_block(uint32_t n_segments);
// positions the padded orders of the pruned population sweeps add (dc_mfma_kernels.hpp: components padded to whole
// query groups: kMaxComp x kMaxGroupRows)
constexpr size_t kOrderPadRows = 64 * 512;
constexpr size_t kMinEdgeMaxRows = ((size_t)1 << 24) - kOrderPadRows;   // (the sweep's deferred-evaluation queue holds 24-bit positions)
// one Boruvka round on the radius graph (d2 < r2): for every component (d_comp[frame] = its id, any
// frame id) the lightest pair that leaves it, by (max(rank), min(rank)) with d_rank[frame] a
// permutation; d_best[id] = (max << 32 | min) or ~0.  d_pops: [n_rows] scratch (populations).
void launch_radius_min_edge(const float* d_coords, uint32_t n_rows, uint32_t n_cols, float r2,
                            const uint32_t* d_comp, const uint32_t* d_rank, unsigned long long* d_best,
                            uint32_t* d_pops, void* d_ws, hipStream_t stream, uint32_t segment = 0,
                            uint32_t n_segments = 0);   // (n_segments > 0: the pairs seen from one segment's queries)
// neighbour sweep over (cell, free energy)-ordered frames with ring-wise pruning
// reuse_components: the component partition an earlier sweep over the same coordinates left in the workspace serves
// this one too (the second call of a populations -> neighbours pair); checked against a cookie on the device
void launch_nn_pruned(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                      uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                      uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws, hipStream_t stream,
                      bool reuse_components = false);
// the same for one SEGMENT of the spatial order (segment s of n: a run of whole query groups): the
// rows a rank of a spatially sharded multi-GPU run answers for.  Outputs as for a row range: zeros /
// "none" for the rows of other segments.
void launch_pop_pruned_segment(const float* d_coords, uint32_t n_rows, uint32_t n_cols,
                               uint32_t segment, uint32_t n_segments, const Rad2& rad2, int n_rad,
                               uint32_t* d_pops_first_row, void* d_ws, hipStream_t stream, bool comp_clean = false);
void launch_nn_pruned_segment(const float* d_coords, uint32_t n_rows, uint32_t n_cols,
                              const float* d_fe, uint32_t segment, uint32_t n_segments,
                              uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2,
                              void* d_ws, hipStream_t stream, bool reuse_components = false);
void launch_nn_mfma(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                    uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                    uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws, hipStream_t stream);

// The rows of one segment of a sharded neighbour sweep compacted into a dense block [4][nn_block_rows] (nn_idx,
// nn_d2 bits, hd_idx, hd_d2 bits by local position), and the gathered blocks of all segments back to the four
// arrays by frame.  pruned: the segment sweep that ran was the pruned matrix-core sweep (its ordering is still in the
// workspace); otherwise, and for flagged data, the segments are the reference's row blocks.
size_t nn_block_rows(size_t n_rows, size_t n_cols, size_t n_segments);
void launch_nn_block_pack(const uint32_t* d_nn_idx, const float* d_nn_d2, const uint32_t* d_hd_idx,
                          const float* d_hd_d2, uint32_t n_rows, uint32_t n_cols, uint32_t segment,
                          uint32_t n_segments, bool pruned, const void* d_ws, uint32_t* d_block, hipStream_t stream);
void launch_nn_block_unpack(const uint32_t* d_blocks, uint32_t n_rows, uint32_t n_cols, uint32_t n_segments,
                            bool pruned, const void* d_ws, uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                            float* d_hd_d2, hipStream_t stream);

// the frame pairs between ADJACENT components of a pruned population sweep (dc_mfma_kernels.hpp "components"), counted
// exactly and added to out[rr * stride + (query position | frame)]
void launch_pop_cross(const float* coords, uint32_t n_cols, const float* coords_r, const uint32_t* perm_r,
                      const float4* box_r, const uint32_t* perm_q, const float4* box_q, const uint32_t* tile_comp_q,
                      const uint32_t* comp, uint32_t T_q, uint32_t group_tiles, QSeg q_seg, int q_in_ref_order,
                      const Rad2& rad2, int n_rad, const uint32_t* hdr, uint32_t* out, size_t stride, int by_position,
                      hipStream_t s);

// diagnostics of the last pruned population sweep that ran in a workspace (synchronises): number of components, the
// global max |x - mean|^2, the bound of max |x - origin(component)|^2 the scale was chosen for, and that scale S
int components_info(const void* d_ws, size_t n_rows, size_t n_cols, uint32_t* n_comp, float* m_global, float* m_local,
                    float* scale, hipStream_t stream);

// Optional timing of the MAIN sweep kernels (bench.py's roofline entry wants the kernel's own duration, not the call's:
// the orderings and operand images are "prep").  When enabled the launchers bracket every main-kernel launch with HIP
// events on the launch stream: the first launch of a kind (0 population, 1 neighbour) since the last read sets the
// start, every launch moves the end.  sweep_timer_read synchronises on the end event.
void sweep_timer_enable(bool on);
void sweep_timer_mark(int kind, bool begin, hipStream_t s);
int sweep_timer_read(int kind, float* ms);   // 0 ok, -1 nothing recorded / error

// dc_sort.hip: stable key/value radix sort on the low key_bits bits of the keys (8 bits per pass).
// remap: the LAST pass writes the values only, into a padded order: the sorted list is cut into n_seg segments
// [seg_start[s], seg_start[s + 1]) (device arrays of n_seg + 1 entries; seg_start[n_seg] = n), segment s lands at
// positions seg_base[s] + 0, 1, ...; tags[position / 32] = s for every tile whose first position is taken.  Positions
// nothing lands on keep their contents (the caller presets them); keys_out is not written then.
constexpr uint32_t kSortMaxSegments = 64;
struct SortRemap {
  const uint32_t* seg_start;
  const uint32_t* seg_base;
  uint32_t n_seg;
  uint32_t* tags;
};
size_t sort_temp_bytes(size_t n);
// keys_in / vals_in are scratch: sorts of three or more passes use them for the passes in between.
int sort_pairs_u32(uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_in,
                   uint32_t* vals_out, size_t n, void* temp, size_t temp_bytes, hipStream_t stream,
                   unsigned key_bits = 32, const SortRemap* remap = nullptr);
constexpr unsigned kCellKeyBits = 24;   // cell keys of the spatial orderings: < 4002^2 < 2^24

}  // namespace dc
