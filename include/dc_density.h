/*
 * dc_density.h -- C ABI of the MI355X-native `clustering density` hot path.
 *
 * Drop-in boundary for moldyn/Clustering's GPU plug-in surface.  Every entry
 * point names the reference interface it replaces (paths relative to the
 * reference's src/).  The reference selects its GPU path at compile time
 * (-DUSE_CUDA) and calls the free functions of density_clustering_cuda.hpp; the
 * C++ shim in clustering_amd/csrc/density_clustering_hip.{hpp,cpp} re-creates
 * exactly those signatures (namespace Clustering::Density::CUDA) on top of this
 * C ABI.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - plain C types only; no C++/torch types cross this boundary.
 *   - every function returns 0 on success or a negative dc_status; the message
 *     is available from dc_hip_last_error() (thread-local).  Nothing here prints
 *     or exits -- the C++ shim restores the reference's "message on stderr, then
 *     exit(EXIT_FAILURE)" convention (density_clustering_cuda.cu:21-30).
 *   - coords: row-major float32 [n_rows][n_cols], as read_coords() produces
 *     (tools.hxx:39-111).  Frame ids are 0-based.
 *   - populations are uint32 on the device like the reference's
 *     (density_clustering_cuda.cu:64,120), laid out radius-major
 *     pops[r*n_rows + i] (density_clustering_cuda.cu:130) IN THE ORDER OF THE
 *     radii ARGUMENT; rows outside [i_from, i_to) are written 0, so that
 *     per-device partials merge by summation (density_clustering_cuda.cu:171-180).
 *   - results follow the reference's OpenMP CPU path bit for bit (the parity
 *     target named by BASELINE.json; density_clustering.cpp:126-288):
 *       pop_r[i] = 1 + #{ j != i : d2(i,j) <  fl32(r*r) }      (strict)
 *       nn[i]    = lexicographic min over j != i of (d2(i,j), j)
 *       nn_hd[i] = same over { j : fe[j] < fe[i] }; empty -> (n_rows+1, FLT_MAX)
 *     with d2 in the reference binary's float summation order.
 *   - "_dev" functions take DEVICE pointers, run on the given hipStream_t
 *     (passed as void*; NULL = the default stream) and are asynchronous unless
 *     stated otherwise.  Functions without the suffix take HOST pointers, manage
 *     device memory themselves and are synchronous -- they mirror the
 *     reference's per-GPU host functions.
 */
#ifndef DC_DENSITY_H
#define DC_DENSITY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define DC_API __attribute__((visibility("default")))
#else
#define DC_API
#endif

typedef enum dc_status {
  DC_OK = 0,
  DC_ERR_INVALID_ARGUMENT = -1,
  DC_ERR_NO_DEVICE = -2,      /* reference: "no CUDA-compatible GPUs found", cuda.cu:37-40 */
  DC_ERR_HIP = -3,            /* a HIP runtime call failed; see dc_hip_last_error() */
  DC_ERR_TOO_LARGE = -4,      /* n_rows + 1 does not fit the uint32 index type */
  DC_ERR_WORKSPACE = -5       /* workspace missing or too small */
} dc_status;

/* kernel family selector for the two pairwise sweeps */
typedef enum dc_variant {
  DC_VARIANT_AUTO = 0,        /* fastest available: pruned MFMA when n_cols allows, else direct */
  DC_VARIANT_DIRECT = 1,      /* VALU, direct differences in the canonical order: exact by construction */
  DC_VARIANT_MFMA = 2,        /* matrix-core Gram form (two fp16 pieces per coordinate, n_cols <= 64) as a
                                 classifier + guard band + canonical re-check; every pair evaluated */
  DC_VARIANT_MFMA_PRUNED = 3, /* the same on spatially ordered frames, skipping tile pairs farther apart than
                                 the radius (the GPU counterpart of the reference's box grid,
                                 density_clustering.cpp:41-89); identical results */
  DC_VARIANT_MFMA32 = 4       /* the literal fp32-input MFMA (v_mfma_f32_32x32x2_f32) Gram form, every pair evaluated:
                                 the instance BASELINE's "fraction of the fp32 MFMA roofline" is quoted on; n_cols
                                 9..10 only, one radius per sweep; identical results (classifier + band + re-check) */
} dc_variant;
/* may be OR-ed into the `variant` argument of the _dev sweeps: the column means, max |x - mean|^2, the
 * non-finite flag and the bounding box in this workspace's header were computed by an earlier sweep over
 * the SAME d_coords (same contents) and are still valid -- the sweep skips its three statistics passes.
 * The reference runs populations and then neighbours over one coordinate array
 * (density_clustering.cpp:616-621, 659-663): the second call of such a pair may set it.  The claim is
 * CHECKED on the device: address, shape and a 64-bit content fingerprint of the whole array (one streaming
 * pass, recomputed by the claiming call) must equal what the statistics pass stored; otherwise -- another
 * array, the same buffer rewritten in place, a reused allocation -- the sweep is answered by the exact
 * direct kernels: slower, same results. */
#define DC_FLAG_STATS_VALID 0x100
#define DC_VARIANT_MASK 0xFF

/* message of the last failing call on this thread ("" if none). */
DC_API const char* dc_hip_last_error(void);

/* library/ABI version, bumped on any change of a signature or of a documented contract.  Hosts check it at
 * load time (clustering_amd/capi.py, density_clustering_hip.cpp).
 *   1  rounds 1-2 up to the sessions
 *   2  sessions (dc_hip_session_*), segment entry points whose populations are PARTIAL counts of ALL rows
 *      (symmetric one-radius sweeps credit both frames of a pair), dc_hip_session_merge_mode, host-merge
 *      fallback, DC_FLAG_STATS_VALID
 *   3  dc_hip_session_merge_note (which merge a multi-device session runs, and why),
 *      dc_hip_workspace_mfma_counters_dev, dc_hip_workspace_layout_status_dev
 *   4  dc_hip_build_digest (which sources this binary was built from)
 *   5  dc_hip_canon_order (which summation order of the reference's distance loop this binary reproduces) */
#define DC_HIP_ABI_VERSION 5
DC_API int dc_hip_abi_version(void);

/* digest of the sources this library was built from (clustering_amd/csrc + include/, comments left out:
 * clustering_amd/csrc/digest.py; 16 hex digits, embedded by the Makefile).  What a measurement is tied to: bench.py
 * prints it as "library_digest" and refuses counter profiles (profiles/ *_pmc.json) taken on another one. */
DC_API const char* dc_hip_build_digest(void);

/* The summation order of the canonical squared distance this library was built for: "sse2" -- the reference's DEFAULT
 * build (density_clustering.cpp:171-176, 263-268 under CMakeLists.txt:37-45: four lane sums, a pair tail) -- or "avx" --
 * a reference built with -DCPU_ACCELERATION=AVX (CMakeLists.txt:73-76: eight lane sums, a four-column step, scalar tails).
 * Integer results are bit-exact against the reference build of the SAME order; the two orders differ in the last bit of
 * some distances, i.e. in a few frames on a radius.  `make -C clustering_amd/csrc CANON=avx` builds the other library
 * into clustering_amd/lib_avx/ (same file name, same ABI). */
DC_API const char* dc_hip_canon_order(void);

/* replaces Clustering::Density::CUDA::get_num_gpus() (density_clustering_cuda.hpp:16-17,
 * density_clustering_cuda.cu:32-43).  Returns the device count (>= 0) or a negative status;
 * 0 devices is NOT an error here (the C++ shim turns it into the reference's exit). */
DC_API int dc_hip_device_count(void);

/* ---------------------------------------------------------------------------------------
 * device-pointer entry points (inputs already resident in HBM)
 * ------------------------------------------------------------------------------------- */

/* bytes of scratch the _dev sweeps may need for this problem size (MFMA operand images,
 * norms; 0-filled by the callee as needed).  Pass a buffer at least this large. */
DC_API size_t dc_hip_workspace_bytes(size_t n_rows, size_t n_cols, size_t n_radii);

/* statistics of the sweeps that last ran in this workspace (asynchronous pruned variants leave them
 * in the workspace header): number of 32x32 frame-pair tiles actually evaluated by the population
 * and by the neighbour sweep (every pair = n_rows^2/1024 per full sweep; the direct kernels and
 * DC_VARIANT_MFMA do not count and report 0).  Synchronises the stream. */
DC_API int dc_hip_workspace_counters_dev(const void* d_workspace, uint64_t* pop_tiles,
                                         uint64_t* nn_tiles, void* stream);
/* ... and the matrix-core instructions (v_mfma_f32_32x32x16_f16: 32*32*16*2 flop each) those sweeps ISSUED, counted by the
 * kernels themselves: tiles x MFMAs per chain for the population sweeps, fewer for the neighbour sweeps, whose chains
 * mostly stop after their coarse MFMAs (the executed-flop figure of bench.py's roofline; it equals
 * SQ_VALU_MFMA_BUSY_CYCLES / 32 of a counter profile).  Synchronises the stream. */
DC_API int dc_hip_workspace_mfma_counters_dev(const void* d_workspace, uint64_t* pop_mfma, uint64_t* nn_mfma,
                                              void* stream);

/* diagnostics of the last pruned POPULATION sweep that ran in this workspace (same n_rows, n_cols): the number of
 * components the frames were cut into (sets at least r_max apart in columns 0/1, each measured from its own origin by
 * the matrix-core sweep; 1 = one origin, the column means), the global max |x - mean|^2, the bound of
 * max |x - origin(component of x)|^2 that the sweep's guard band follows, and the scale S of that sweep (band = 1 / S in
 * the units of d2).  Synchronises the stream. */
DC_API int dc_hip_workspace_components_dev(const void* d_workspace, size_t n_rows, size_t n_cols,
                                          uint32_t* n_components, float* extent2_global, float* extent2_local,
                                          float* scale, void* stream);

/* measurement aid (bench.py's roofline entry): with timing enabled the library brackets its MAIN sweep
 * kernels (population_count / nearest_neighbor_search counterparts) with HIP events on the launch stream;
 * dc_hip_last_sweep_ms returns the duration of the kernels of one kind (0 population, 1 neighbour) launched on
 * the calling thread's current device since the previous read -- the sweep kernel alone, without the ordering /
 * operand-image passes of the call.  Synchronises on the kernel's end.  Returns DC_ERR_INVALID_ARGUMENT if no
 * sweep of that kind was recorded. */
DC_API int dc_hip_sweep_timing(int enable);
DC_API int dc_hip_last_sweep_ms(int kind, float* ms);

/* replaces the kernel loop of calculate_populations_per_gpu (density_clustering_cuda.cu:45-137;
 * kernel population_count, density_clustering_cuda_kernels.cu:9-56): ONE launch sweeps all
 * n_rows reference frames for the query rows [i_from, i_to).
 *   d_coords  [n_rows*n_cols] float32, device
 *   radii     [n_radii] float32, HOST (tiny; the reference also passes them from the host)
 *   d_pops    [n_radii*n_rows] uint32, device, fully overwritten (0 outside the row range) */
DC_API int dc_hip_populations_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                  const float* radii, size_t n_radii, size_t i_from, size_t i_to,
                                  uint32_t* d_pops, void* d_workspace, size_t workspace_bytes,
                                  int variant, void* stream);

/* The same sweep for one SEGMENT of a sharded run instead of a row range: rank `segment` of
 * `n_segments` (density_clustering_cuda.cu:149-169 gives every GPU a contiguous block of rows; any
 * partition serves, as long as the partial results merge).  With the pruned matrix-core sweep a
 * segment is every n_segments-th query group of the SPATIAL order (128 - 192 frames of one grid cell):
 * a rank's queries are as compact as those of a full sweep and prune as well (a block of consecutive
 * rows of a trajectory is spread over the whole conformational space), and dealing the groups out
 * cyclically gives every rank the same mix of dense and sparse regions; with every other variant,
 * n_cols > 64 or non-finite data the
 * segment is the reference's row block.  d_pops: PARTIAL populations that merge by summation over the
 * segments (the reference merges its per-GPU partials the same way, density_clustering_cuda.cu:171-180).
 * A one-radius pruned sweep evaluates every pair of query groups once and credits both frames
 * (d2(i,j) = d2(j,i), the reference's own i < j loop, density_clustering.cpp:170-182), so a segment's
 * counts cover all rows; the other sweeps write the final counts of the segment's own rows and zeros
 * elsewhere.  Only the sum over all segments is a population. */
DC_API int dc_hip_populations_segment_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                          const float* radii, size_t n_radii, size_t segment,
                                          size_t n_segments, uint32_t* d_pops, void* d_workspace,
                                          size_t workspace_bytes, int variant, void* stream);

/* replaces Clustering::Density::calculate_free_energies (density_clustering.cpp:197-212), which the
 * reference runs on the host in both builds.  fe[i] = (float)-log((double)((float)pop[i] * (1.0f/max)))
 * -- with the reference's bits: the device evaluates the double log, and every row whose value lies
 * within 64 ulp(double) of a float rounding boundary (where a few ulp of difference between two libms
 * could show) is recomputed by the HOST libm.  Synchronises the stream once (reads max_pop).
 *   d_pops [n_rows] uint32 device (one radius), d_fe [n_rows] float32 device.
 * max_pop_out (optional, host) receives the maximum population. */
DC_API int dc_hip_free_energies_dev(const uint32_t* d_pops, size_t n_rows, float* d_fe,
                                    uint32_t* max_pop_out, void* stream);

/* replaces the kernel loop of nearest_neighbors_per_gpu (density_clustering_cuda.cu:184-284; kernel
 * nearest_neighbor_search, density_clustering_cuda_kernels.cu:58-130) for query rows [i_from, i_to).
 *   d_fe      [n_rows] float32 device
 *   d_nn_idx / d_hd_idx [n_rows] uint32, d_nn_d2 / d_hd_d2 [n_rows] float32, device; rows outside the
 *   range are written with the "none" value (n_rows+1, FLT_MAX) of density_clustering.cpp:242-245. */
DC_API int dc_hip_nearest_neighbors_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                        const float* d_fe, size_t i_from, size_t i_to,
                                        uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                                        float* d_hd_d2, void* d_workspace, size_t workspace_bytes,
                                        int variant, void* stream);

/* neighbour sweep for one segment of a sharded run (see dc_hip_populations_segment_dev); rows of other
 * segments hold (n_rows+1, FLT_MAX), so partials merge by taking, per row, the minimum of
 * (d2 bits << 32 | index) -- only the owner's value is smaller than the "none" value. */
DC_API int dc_hip_nearest_neighbors_segment_dev(const float* d_coords, size_t n_rows, size_t n_cols,
                                                const float* d_fe, size_t segment, size_t n_segments,
                                                uint32_t* d_nn_idx, float* d_nn_d2,
                                                uint32_t* d_hd_idx, float* d_hd_d2,
                                                void* d_workspace, size_t workspace_bytes,
                                                int variant, void* stream);

/* Merging the neighbour partials of a sharded run (density_clustering_cuda.cu:311-326 overwrites row by
 * row on the host): [2][n_rows] order-preserving 64-bit words (d2 bits << 32 | index) -- nn first, then
 * nn_hd.  d2 >= 0, so the words of a row order like the lexicographic (d2, index): the owner's word is
 * the minimum over the ranks (everybody else holds the "none" value (n_rows+1, FLT_MAX)), i.e. ONE
 * all-reduce(min) of int64 merges all four arrays. */
DC_API int dc_hip_neighbors_pack_dev(const uint32_t* d_nn_idx, const float* d_nn_d2,
                                     const uint32_t* d_hd_idx, const float* d_hd_d2, size_t n_rows,
                                     unsigned long long* d_words, void* stream);
DC_API int dc_hip_neighbors_unpack_dev(const unsigned long long* d_words, size_t n_rows,
                                       uint32_t* d_nn_idx, float* d_nn_d2, uint32_t* d_hd_idx,
                                       float* d_hd_d2, void* stream);

/* The same merge as an ALL-GATHER (half the bytes, no reduction): every rank compacts the results of its own
 * segment's rows into a dense block [4][block_rows] uint32 (nn_idx, nn_d2 bits, hd_idx, hd_d2 bits, by local
 * position in the segment; pad entries hold the "none" value), the blocks of all ranks are gathered into
 * [n_segments][4][block_rows], and the gathered blocks are scattered back to the four arrays by frame.
 * With the pruned matrix-core sweep a segment is every n_segments-th query group of the sweep's spatial order,
 * which every rank derives identically from the replicated coordinates and free energies: pack and unpack read
 * the ordering that dc_hip_nearest_neighbors_segment_dev left in THIS rank's workspace (same n_rows, n_cols,
 * variant; no other sweep in between).  Otherwise (other variants, n_cols > 64, flagged data) the blocks are the
 * reference's row blocks (density_clustering_cuda.cu:293, 305-308, 311-326). */
DC_API size_t dc_hip_neighbors_block_rows(size_t n_rows, size_t n_cols, size_t n_segments);
DC_API int dc_hip_neighbors_block_pack_dev(const uint32_t* d_nn_idx, const float* d_nn_d2,
                                           const uint32_t* d_hd_idx, const float* d_hd_d2, size_t n_rows,
                                           size_t n_cols, size_t segment, size_t n_segments,
                                           const void* d_workspace, size_t workspace_bytes, int variant,
                                           uint32_t* d_block, void* stream);
DC_API int dc_hip_neighbors_block_unpack_dev(const uint32_t* d_blocks, size_t n_rows, size_t n_cols,
                                             size_t n_segments, const void* d_workspace,
                                             size_t workspace_bytes, int variant, uint32_t* d_nn_idx,
                                             float* d_nn_d2, uint32_t* d_hd_idx, float* d_hd_d2, void* stream);
/* The unpack compares the layout headers of the gathered blocks ON THE DEVICE before it scatters anything: on a mismatch
 * (a rank derived another order) NOTHING is written -- the output arrays keep whatever they held -- and a word of the
 * workspace header records the verdict of THAT unpack (pruned variants only; with any other variant the workspace is not
 * touched and the host compares the headers itself).  The word describes the LAST unpack only: the next unpack, and the
 * preparation of the next sweep in the workspace, overwrite it.  This call reads it (synchronises the stream): *mismatch =
 * 1 if the last unpack refused its blocks.  A host that wants every step covered asks after every step, before the next
 * call into the workspace (clustering_amd.distributed does unless told otherwise; bench.py asks after its instrumented
 * steps and after the last timed one). */
DC_API int dc_hip_workspace_layout_status_dev(const void* d_workspace, int* mismatch, void* stream);

/* replaces Clustering::Density::compute_sigma2 (density_clustering.cpp:334-343): mean of the nearest-
 * neighbour d2 accumulated in double IN FRAME ORDER (bit-stable), on the device (one block, fixed
 * reduction tree would change bits -- so this is a single ordered pass over a device->host copy).
 * Synchronises the stream. */
DC_API int dc_hip_sigma2_dev(const float* d_nn_d2, size_t n_rows, double* sigma2_out, void* stream);

/* the radius graph that the reference's screening walks one frame at a time: replaces the O(n^2) scans
 * of high_density_neighborhood (density_clustering.cpp:292-332, called per frame from
 * density_clustering_common.cpp:97-121; CUDA: density_clustering_cuda.cu:396-594) by ONE pruned sweep
 * that lists every unordered frame pair {i, j}, i != j, whose canonical d2 is < r2 (strict, :319).
 *   r2       the squared distance itself (the reference passes max_dist = 4*sigma2, a float)
 *   d_pops   [n_rows] uint32 device, out: populations at that radius (1 + number of partners)
 *   d_pairs  [capacity][2] uint32 device, out: frame ids of the pairs, in no particular order, each
 *            pair once; may be NULL with capacity 0 to count only
 *   d_count  device uint64, out: number of pairs found -- if it exceeds capacity only the first
 *            `capacity` were written (call again with a larger buffer); ~0 if the coordinates are not
 *            finite (no pairs are produced then).  Needs n_cols <= 64 and a workspace as above. */
DC_API int dc_hip_radius_pairs_dev(const float* d_coords, size_t n_rows, size_t n_cols, float r2,
                                   uint32_t* d_pops, uint32_t* d_pairs, size_t capacity,
                                   unsigned long long* d_count, void* d_workspace,
                                   size_t workspace_bytes, void* stream);

/* one Boruvka round on that radius graph, for screenings that only need its CONNECTIVITY below a
 * free-energy threshold (density_clustering_common.cpp:37-134 started from an empty clustering): for
 * every component the lightest pair that leaves it.  No pair list is materialised.
 *   d_comp   [n_rows] uint32 device: component id of every frame (the id is any frame id < n_rows)
 *   d_rank   [n_rows] uint32 device: a permutation of 0..n_rows-1 (position in order of free energy);
 *            the weight of a pair is (max(rank), min(rank)), compared lexicographically
 *   d_best   [n_rows] uint64 device, out: d_best[id] = (max << 32 | min) of the lightest pair with
 *            canonical d2 < r2 that joins component id to another one, ~0 if there is none
 *   d_pops   [n_rows] uint32 device, out: populations at that radius
 * Needs n_cols <= 64, n_rows <= 2^24 and a workspace as above; with coordinates that are not finite
 * every d_best entry stays ~0 (dc_hip_radius_forest reports the error). */
DC_API int dc_hip_radius_min_edge_dev(const float* d_coords, size_t n_rows, size_t n_cols, float r2,
                                      const uint32_t* d_comp, const uint32_t* d_rank,
                                      unsigned long long* d_best, uint32_t* d_pops, void* d_workspace,
                                      size_t workspace_bytes, void* stream);
/* the same for one segment of a sharded run (n_segments > 0): only the pairs seen from the queries of
 * that segment enter d_best (every pair is seen from both of its ends, possibly by different segments);
 * the partial d_best arrays merge by an element-wise UNSIGNED minimum, the partial d_pops by summation. */
DC_API int dc_hip_radius_min_edge_segment_dev(const float* d_coords, size_t n_rows, size_t n_cols, float r2,
                                              const uint32_t* d_comp, const uint32_t* d_rank,
                                              size_t segment, size_t n_segments,
                                              unsigned long long* d_best, uint32_t* d_pops,
                                              void* d_workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * host-pointer entry points (mirror the reference's per-GPU host functions)
 * ------------------------------------------------------------------------------------- */

/* replaces calculate_populations_per_gpu(coords, n_rows, n_cols, radii, i_from, i_to, i_gpu)
 * (density_clustering_cuda.cu:45-52; declared as calculate_populations_partial in
 * density_clustering_cuda.hpp:21-30).  pops: HOST [n_radii*n_rows] uint32. */
DC_API int dc_hip_populations(const float* coords, size_t n_rows, size_t n_cols, const float* radii,
                              size_t n_radii, size_t i_from, size_t i_to, int device, uint32_t* pops);

/* replaces nearest_neighbors_per_gpu(coords, n_rows, n_cols, free_energy, i_from, i_to, i_gpu)
 * (density_clustering_cuda.cu:184-191).  All outputs HOST [n_rows]. */
DC_API int dc_hip_nearest_neighbors(const float* coords, size_t n_rows, size_t n_cols, const float* fe,
                                    size_t i_from, size_t i_to, int device, uint32_t* nn_idx,
                                    float* nn_d2, uint32_t* hd_idx, float* hd_d2);

/* host-pointer form of dc_hip_radius_pairs_dev.  pairs: HOST [capacity][2]; *count: pairs found (if
 * larger than capacity, call again with a buffer of that size). */
DC_API int dc_hip_radius_pairs(const float* coords, size_t n_rows, size_t n_cols, float r2, int device,
                               uint32_t* pairs, size_t capacity, unsigned long long* count);

/* bottleneck spanning forest of the radius graph: Boruvka rounds of dc_hip_radius_min_edge_dev with
 * the components merged on the host.  For every t the pairs of the forest with max(rank) < t connect
 * exactly the frames that the pairs of the full graph with max(rank) < t connect -- what a screening
 * threshold asks for -- with at most n_rows-1 pairs instead of all of them (9e8 at C3).
 *   rank     HOST [n_rows]: permutation of 0..n_rows-1
 *   edges    HOST [n_rows-1][2] uint32, out: frame ids of the forest's pairs
 *   n_edges  out: number of pairs written;  n_rounds (may be NULL): sweeps that were run */
DC_API int dc_hip_radius_forest(const float* coords, size_t n_rows, size_t n_cols, float r2,
                                const uint32_t* rank, int device, uint32_t* edges, size_t* n_edges,
                                uint32_t* n_rounds);

/* whole path on n_devices GPUs of this process (devices 0..n_devices-1; <= 0: all): ONE session (below)
 * -- open, populations, free energies of radius index fe_radius_index, neighbours, close.  Replaces the
 * pair CUDA::calculate_populations / CUDA::nearest_neighbors (density_clustering_cuda.cu:139-182,
 * :286-328), which upload the coordinates twice and merge every partial result on the host.
 * pops HOST [n_radii*n_rows]; fe, nn_*, hd_* HOST [n_rows] (fe may be NULL if nn_idx is; NN skipped if
 * nn_idx == NULL). */
DC_API int dc_hip_density_all(const float* coords, size_t n_rows, size_t n_cols, const float* radii,
                              size_t n_radii, size_t fe_radius_index, int n_devices, uint32_t* pops,
                              float* fe, uint32_t* nn_idx, float* nn_d2, uint32_t* hd_idx,
                              float* hd_d2);

/* ---------------------------------------------------------------------------------------
 * sessions: a trajectory resident on the GPUs across the phases of density_clustering.cpp:597-817
 * ------------------------------------------------------------------------------------- */
/* The reference's GPU host code allocates, uploads the coordinates and frees again inside every call
 * (density_clustering_cuda.cu:65-81, :133-135, :201-225, :278-281) and merges per-GPU partials on the
 * host (:171-180, :311-326).  A session uploads once per device and keeps coordinates, operand
 * workspace, populations, free energies and neighbours in HBM; one host thread drives each device
 * (like :152-157, :295-299); with more than one device the partial results merge ON the devices with
 * RCCL over xGMI -- all-reduce(sum, uint32) of the [n_radii][n_rows] populations, all-reduce(min,
 * uint64) of the packed neighbour words and of the forest's candidates -- and only final arrays cross
 * PCIe, from device 0.  Every device answers for one segment (dc_hip_*_segment_dev).  Host output
 * pointers may be NULL (the result then only stays resident for the next phase).  A session is used
 * by one host thread at a time. */
typedef struct dc_hip_session dc_hip_session;

/* devices: n_devices device ordinals, or NULL for 0..n_devices-1; n_devices <= 0: all devices.
 * coords: HOST, read during the call only.  RCCL (librccl.so.1) is loaded on first need, i.e. when a
 * session spans more than one device; if it cannot be loaded or its communicator cannot be built the
 * session merges its partials through the HOST instead, exactly like the reference
 * (density_clustering_cuda.cu:171-180, :311-326): same results, PCIe instead of xGMI.
 * Environment: DC_SESSION_MERGE=host forces the host merge, =rccl makes a missing RCCL an error.
 * The calling thread's current device is restored by every session entry point. */
DC_API int dc_hip_session_open(const float* coords, size_t n_rows, size_t n_cols, const int* devices,
                               int n_devices, dc_hip_session** session);
DC_API void dc_hip_session_close(dc_hip_session* session);
DC_API int dc_hip_session_devices(const dc_hip_session* session);      /* number of devices */
DC_API int dc_hip_session_uses_rccl(const dc_hip_session* session);    /* 1 if partials merge over RCCL */
/* 0: one device, nothing to merge; 1: RCCL collectives on the devices; 2: through the host */
DC_API int dc_hip_session_merge_mode(const dc_hip_session* session);
/* one line saying which merge this session runs and, for the host merge of a multi-device session, WHY RCCL is not
 * used (library not loadable, communicator not built, DC_SESSION_MERGE=host, duplicate devices) -- the reference's
 * merge (density_clustering_cuda.cu:152-180) is never silent about what it does, and a multi-GPU run that fell back
 * to PCIe is correct and slow: the C++ shim prints this line to stderr for mode 2, the command line under -v.
 * The string lives as long as the session. */
DC_API const char* dc_hip_session_merge_note(const dc_hip_session* session);
/* 32x32 frame-pair tiles the last population call / neighbour call evaluated, summed over devices */
DC_API int dc_hip_session_counters(const dc_hip_session* session, uint64_t* pop_tiles, uint64_t* nn_tiles);

/* CUDA::calculate_populations (density_clustering_cuda.cu:139-182) on the resident coordinates:
 * populations for n_radii radii, in the order given.  pops: HOST [n_radii*n_rows] or NULL. */
DC_API int dc_hip_session_populations(dc_hip_session* session, const float* radii, size_t n_radii,
                                      uint32_t* pops);
/* calculate_free_energies (density_clustering.cpp:197-212) of the resident populations of radius index
 * radius_index, on every device.  fe: HOST [n_rows] or NULL; max_pop: optional. */
DC_API int dc_hip_session_free_energies(dc_hip_session* session, size_t radius_index, float* fe,
                                        uint32_t* max_pop);
/* free energies from the caller instead (-D re-use, density_clustering.cpp:600-611; and the reference's
 * nearest_neighbors(coords, ..., free_energy) signature).  fe: HOST [n_rows]. */
DC_API int dc_hip_session_set_free_energies(dc_hip_session* session, const float* fe);
/* CUDA::nearest_neighbors (density_clustering_cuda.cu:286-328) from the resident free energies, and
 * compute_sigma2 (density_clustering.cpp:334-343: double sum in frame order).  All outputs HOST
 * [n_rows] or NULL; sigma2 optional. */
DC_API int dc_hip_session_nearest_neighbors(dc_hip_session* session, uint32_t* nn_idx, float* nn_d2,
                                            uint32_t* hd_idx, float* hd_d2, double* sigma2);
/* dc_hip_radius_pairs on the resident coordinates (device 0 of the session). */
DC_API int dc_hip_session_radius_pairs(dc_hip_session* session, float r2, uint32_t* pairs, size_t capacity,
                                       unsigned long long* count);
/* dc_hip_radius_forest on the resident coordinates; with several devices every Boruvka round is one
 * segment sweep per device + one all-reduce(min, uint64) of the per-component candidates. */
DC_API int dc_hip_session_radius_forest(dc_hip_session* session, float r2, const uint32_t* rank,
                                        uint32_t* edges, size_t* n_edges, uint32_t* n_rounds);

#ifdef __cplusplus
}
#endif
#endif /* DC_DENSITY_H */
