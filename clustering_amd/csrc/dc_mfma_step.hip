// dc_mfma_step.hip -- instantiates the MFMA sweeps for ONE MFMA count per tile pair (compile with
// -DDC_STEP=n, n = nm_for(n_cols) = ceil((3 n_cols + 2) / 16): three piece products per column and two constant slots on
// the K axis, 16 slots per MFMA); see dc_mfma_kernels.hpp.
#include "dc_mfma_kernels.hpp"

#ifndef DC_STEP
#error "compile with -DDC_STEP=<MFMAs per tile pair>"
#endif
#define DC_CAT2(a, b) a##b
#define DC_CAT(a, b) DC_CAT2(a, b)

namespace dc {

void DC_CAT(pop_mfma_step_, DC_STEP)(const float* coords, uint32_t n_rows, uint32_t n_cols,
                                     void* d_ws, uint32_t i_from, uint32_t i_to, const Rad2& rad2,
                                     int n_rad, uint32_t* pops, hipStream_t s) {
  const Layout L = make_layout(n_rows, n_cols);
  pop_dispatch<DC_STEP>(coords, n_rows, n_cols, ws_ptrs(d_ws, L), L.T, i_from, i_to, rad2, n_rad,
                        pops, s);
}

void DC_CAT(pop_pruned_step_, DC_STEP)(const float* coords, uint32_t n_rows, uint32_t n_cols,
                                       void* d_ws, uint32_t T_ref, uint32_t n_q, int q_mode, QSeg q_seg,
                                       const Rad2& rad2, int n_rad, uint32_t* pops,
                                       const EdgeSink* sink, hipStream_t s, bool pos_clean) {
  const Layout L = make_layout(n_rows, n_cols);
  // evaluated-chain counter: header word 2..3 (8-byte aligned)
  // (T_ref: tiles of the padded reference order; n_q: positions of the query order)
  pop_pruned_dispatch<DC_STEP>(coords, n_rows, n_cols, ws_ptrs(d_ws, L), T_ref, n_q, q_mode, q_seg,
                               rad2, n_rad, pops, (unsigned long long*)((char*)d_ws + 8), sink, s, pos_clean);
}

void DC_CAT(nn_pruned_step_, DC_STEP)(const float* coords, uint32_t n_rows, uint32_t n_cols,
                                      const float* fe, void* d_ws, uint32_t T_ref, uint32_t n_q, int q_mode,
                                      QSeg q_seg, float cell2, uint32_t* nn_idx, float* nn_d2,
                                      uint32_t* hd_idx, float* hd_d2, hipStream_t s) {
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  NnPrunedArgs A;
  A.img_r = (const uint4*)(p + L.off_img_p);
  A.norms_r = (const float*)(p + L.off_norm_p);
  A.perm_r = (const uint32_t*)(p + L.off_perm_p);
  A.box_r = (const float4*)(p + L.off_box_p);
  A.box_t = (float4*)(p + L.off_box_t);
  A.ferange_r = (const float2*)(p + L.off_ferange_p);
  A.fe_c = (const float*)(p + L.off_fe_s);
  A.coords_c = (const float*)(p + L.off_coords_p);
  A.invpos_r = (const uint32_t*)(p + L.off_invpos);
  // query operands: own ordering of a row range, or the reference order (all rows / a window of it)
  const bool own = q_mode == kQueryOwnOrder;
  A.img_q = (const uint4*)(p + L.off_img_q);   // B form of the query rows
  A.norms_q = own ? (const float*)(p + L.off_norm_q) : A.norms_r;
  A.perm_q = own ? (const uint32_t*)(p + L.off_perm_q) : A.perm_r;
  A.box_q = own ? (const float4*)(p + L.off_box_q) : A.box_r;
  A.q_seg = q_seg;
  A.merge64 = (unsigned long long*)(p + L.off_merge64);
  A.n_q = n_q;
  A.full_range = (q_mode == kQueryAll) ? 1 : 0;   // 0: positions of the queries come from invpos
  A.cell2 = cell2;
  A.tile_comp_q = (const uint32_t*)(p + (own ? L.off_tile_comp_q : L.off_tile_comp));
  A.comp = (const uint32_t*)(p + L.off_comp);
  // (T_ref: tiles of the padded reference order; n_q: positions of the query order)
  nn_pruned_dispatch<DC_STEP>(coords, n_rows, n_cols, fe, A, T_ref, (const uint32_t*)p,
                              (unsigned long long*)(p + 16), nn_idx, nn_d2, hd_idx, hd_d2, s);
}

void DC_CAT(nn_mfma_step_, DC_STEP)(const float* coords, uint32_t n_rows, uint32_t n_cols,
                                    void* d_ws, uint32_t i_from, uint32_t i_to, uint32_t* nn_idx,
                                    float* nn_d2, uint32_t* hd_idx, float* hd_d2, hipStream_t s) {
  const Layout L = make_layout(n_rows, n_cols);
  nn_dispatch<DC_STEP>(coords, n_rows, n_cols, ws_ptrs(d_ws, L), L.T, i_from, i_to, nn_idx, nn_d2,
                       hd_idx, hd_d2, s);
}

}  // namespace dc

