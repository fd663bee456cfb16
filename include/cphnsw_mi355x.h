/* cphnsw_mi355x.h — C-ABI of the MI355X-native CP-HNSW layer-0 hot path.
 *
 * One shared library (libcphnsw_mi355x.so, built by hipcc for gfx950) replaces, for the
 * query-time path only, what the reference binds through pybind11 in
 * src/bindings.cpp:115-240 (class `cphnsw._core.CPIndex`):
 *
 *   reference interface (file:line)                      this header
 *   ---------------------------------------------------  ------------------------------
 *   CPIndex(dim, bits)          src/bindings.cpp:77-123   cph_create
 *   ~CPIndex                                             cph_destroy
 *   .load(path)                 src/bindings.cpp:227-232  cph_load   (v2 file, api/hnsw_index.hpp:305-443)
 *   .save(path)                 src/bindings.cpp:220-225  cph_save   (api/hnsw_index.hpp:217-303)
 *   .search(query,k)            src/bindings.cpp:146-175  cph_search
 *   .search_batch(queries,k)    src/bindings.cpp:177-218  cph_search_batch / cph_search_batch_device
 *   .size/.dim/.is_finalized    src/bindings.cpp:234-239  cph_size / cph_dim / cph_is_finalized
 *   .build/.finalize            src/bindings.cpp:125-144  cph_build / cph_finalize
 *
 * Kernel-level hooks (parity tests and the roofline benchmark; they expose the units the
 * reference implements in distance/fastscan_kernel.hpp, core/memory.hpp and
 * encoder/rabitq_encoder.hpp):
 *   cph_encode_query, cph_entry_point, cph_fastscan_block, cph_exact_l2, cph_fastscan_stream_*.
 *
 * Conventions: plain pointers and sizes only; every function returns a cph_status; on
 * failure cph_last_error() (thread-local) holds the message the reference would have
 * thrown.  CPH_INVALID_ARGUMENT maps to Python ValueError (std::invalid_argument),
 * CPH_RUNTIME_ERROR to RuntimeError (std::runtime_error), CPH_OUT_OF_MEMORY to MemoryError.
 * The caller owns all buffers; the library owns the handle.  A handle is bound to one HIP
 * device.  Threads: every entry point may be called from any thread.  Concurrent cph_search callers on one handle
 * are COALESCED into shared launches (the reference answers them in parallel under a shared lock,
 * src/bindings.cpp:146-175, api/hnsw_index.hpp:172): a caller that finds a free leader slot takes everybody queued
 * so far with the same k into one launch; each waits for its own query only and gets its own rows.  The other entry points serialise on the handle
 * (cph_search_batch_device only while it enqueues).  Knobs: CPH_LEADER_SLOTS (default 3), CPH_GATHER_US (80).
 * Returned ids are the reference's internal (post-BFS-reorder) node ids.
 */
#ifndef CPHNSW_MI355X_H
#define CPHNSW_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cph_index cph_index;

typedef enum {
    CPH_OK = 0,
    CPH_INVALID_ARGUMENT = 1,
    CPH_RUNTIME_ERROR = 2,
    CPH_OUT_OF_MEMORY = 3,
    CPH_NOT_IMPLEMENTED = 4
} cph_status;

/* Thread-local message of the last failing call on this thread. */
const char* cph_last_error(void);

/* Library/ABI version (major*100 + minor). */
int cph_version(void);

/* ---- lifecycle ------------------------------------------------------------------- */
/* dim > 0, next_pow2(dim) in {16..2048}, bits in {1,2,4}; device = HIP device ordinal. */
int cph_create(uint64_t dim, uint64_t bits, int device, cph_index** out);
int cph_destroy(cph_index* h);

int cph_load(cph_index* h, const char* path);
int cph_save(cph_index* h, const char* path);
/* GPU-native index file (csrc/native_file.h): the device block layout, vectors and norms as the GPU reads
 * them, so loading is an mmap plus two host-to-device copies instead of the v2 file's per-vertex
 * re-layout.  A handle loaded this way can still write a v2 file (cph_save) for the reference. */
int cph_save_native(cph_index* h, const char* path);
int cph_load_native(cph_index* h, const char* path);
int cph_size(cph_index* h, uint64_t* n);
int cph_dim(cph_index* h, uint64_t* dim);
int cph_is_finalized(cph_index* h, int* flag);

/* Index construction (api/hnsw_index.hpp:93-166).  build() copies the
 * n x dim float32 vectors; finalize() builds the index on the GPU (exact 32-NN on the matrix cores,
 * reverse edges + neighbour selection, per-edge RaBitQ codes written straight into the search layout,
 * upper layers, calibration sampling; csrc/builder.h).  The edge encoder is bit-exact with the
 * reference's; the graph has statistical parity (the reference's own build depends on its thread count). */
int cph_build(cph_index* h, const float* vectors, uint64_t n);
int cph_finalize(cph_index* h);
/* Construction / ground-truth hook: exact 32 nearest neighbours (squared L2, ascending) by brute
 * force on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32).  queries == NULL: of every row of
 * vectors[n][dim] against the other rows (self excluded) -- the working lists the reference obtains
 * from NNDescent (graph/graph_refinement.hpp:455-515; distances core/memory.hpp:65-79), ids/dist
 * [n][32].  queries != NULL: of queries[nq][dim] against vectors[n][dim], ids/dist [nq][32].  Rows
 * with fewer than 32 candidates are padded with 0xFFFFFFFF / FLT_MAX. */
int cph_knn_bruteforce(int device, const float* vectors, uint64_t n, uint64_t dim, const float* queries,
                       uint64_t nq, uint32_t* ids, float* dist);

/* Construction hook: the data-side encoder of one vertex' edges on the GPU (the kernel finalize() runs
 * for every vertex; encoder/rabitq_encoder.hpp:138-181, 287-323, 371-467): parent and cnt <= 32
 * neighbours (dim floats each) -> values u8[cnt][D] (code value per dimension), aux f32[cnt][3] =
 * {nop, ip_qo, ip_cp}, pops u32[cnt][2] = {msb popcount, weighted popcount}. */
int cph_encode_edges(int device, uint64_t dim, uint64_t bits, const float* parent, const float* nbrs, uint64_t cnt,
                     uint8_t* values, float* aux, uint32_t* pops);

/* Construction hook: the neighbour-selection kernel on ONE vertex with a given candidate list -- the rule of
 * graph/neighbor_selection.hpp:21-88 (select_neighbors_alpha_cng), deterministic on a fixed list.  x = [n][D] padded
 * vectors, fwd = 32 forward candidates (0xFFFFFFFF = none), rev = up to 96 further candidates (the reverse edges),
 * err = per-vertex margin terms or null.  out_ids = 32 selected ids (0xFFFFFFFF padded), *out_cnt = how many. */
int cph_select_hook(int device, const float* x, uint64_t n, uint64_t D, uint32_t vertex, const uint32_t* fwd,
                    const uint32_t* rev, uint64_t n_rev, uint32_t R, float alpha, float tau, float alpha_max,
                    const float* err, uint32_t* out_ids, uint32_t* out_cnt);

/* Construction hook: the calibration sampler (api/hnsw_index.hpp:770-1040 gathers the same quantities) on given sample
 * queries [ns][dim] and start vertices: per sample one greedy hop, then per edge of the vertex arrived at
 * rec[ns][32][6] = {nop, ip_est_raw - ip_cp, max(|ip_qo|, 1e-10), <q - p, o - p> / nop, |q - o|^2, ip_qo};
 * rec_cnt[ns] = valid edges, dqp[ns] = exact |q - p|^2. */
int cph_calib_hook(cph_index* h, const float* queries, const uint32_t* start, uint64_t ns, float* rec, uint32_t* rec_cnt,
                   float* dqp);

/* Self-test hook for the beam's heap routines (wave-parallel std::push_heap / std::pop_heap with the first 255
 * entries in LDS and the rest in HBM, search/rabitq_search.hpp:53-58, :79-80): runs `ops` (1 = push the next
 * (key, id), 0 = pop) on one wave and returns the heap array; the test compares it with libstdc++'s on the same
 * sequence.  out_keys / out_ids hold n_push entries. */
int cph_debug_heap_ops(int device, const uint8_t* ops, uint64_t n_ops, const float* keys, const uint32_t* ids, uint64_t n_push,
                       float* out_keys, uint32_t* out_ids, uint32_t* out_size);

/* ---- search ---------------------------------------------------------------------- */
/* queries: host, row-major [n][dim] float32.  ids/dist: host, [n][k], rows shorter than k
 * padded with -1 / FLT_MAX (src/bindings.cpp:202-210). */
int cph_search_batch(cph_index* h, const float* queries, uint64_t n, uint64_t k,
                     int64_t* ids, float* dist);

/* Same, but queries/ids/dist are DEVICE pointers on the handle's device and the work is
 * enqueued on `stream` (a hipStream_t; NULL = default stream).  Nothing is copied to the host and
 * the call returns after enqueueing, always: a query that outgrows its scratch capacity is
 * answered exactly by a full-capacity re-run launch enqueued behind the main one.  The results are
 * complete when `stream` reaches the point behind this call.  A handle keeps two sets of batch
 * scratch and uses them alternately, so two batches enqueued on two different streams run
 * concurrently (the second fills the GPU while the first drains its longest queries); a third
 * call waits, on its stream, for the batch that used its set before.  d_queries must stay valid
 * until the batch has run. */
int cph_search_batch_device(cph_index* h, const float* d_queries, uint64_t n, uint64_t k,
                            int64_t* d_ids, float* d_dist, void* stream);
/* Blocks the calling host thread until every batch enqueued on this handle has finished. */
int cph_synchronize(cph_index* h);

/* Single query; writes m <= max(k,1) results (unpadded, src/bindings.cpp:146-175). */
int cph_search(cph_index* h, const float* query, uint64_t k, int64_t* ids, float* dist,
               uint64_t* m);

/* Stored vectors of internal ids [first, first+count) (dim floats each, row-major).  The reference
 * never exposes its BFS permutation; a harness recovers internal -> input row numbers by
 * matching these rows against its own base vectors. */
int cph_get_vectors(cph_index* h, uint64_t first, uint64_t count, float* out);

/* Batch scratch sets in rotation (default 2, at most 4): that many batches enqueued on different streams can be in
 * flight together, each on its own slots -- with fewer slots per batch (cph_set_search_params) the drain of one batch
 * is filled by the others.  Waits for everything enqueued on the handle. */
int cph_set_batch_sets(cph_index* h, uint32_t n_sets);

/* Tuning knobs (0 = automatic): resident query slots and per-slot beam capacity. */
int cph_set_search_params(cph_index* h, uint32_t slots, uint64_t beam_capacity);

/* Per-batch work counters of the last search enqueued on this handle (sums over queries; waits
 * for that batch): out[0]=expansions (FastScan blocks), [1]=exact L2 evaluations, [2]=new
 * neighbours, [3]=beam pushes, [4]=stage-2 skipped batches, [5]=queries re-run after a capacity
 * overflow, [6]=device time of the search launches in microseconds (HIP events on the launch
 * stream; with two batches in flight it includes the time shared with the other one),
 * [7]=expansions whose 32 neighbours were all estimated already, [8]=resident query slots used,
 * [9]=per-slot capacity that launch ran with, and for the probe-first instantiations (D = 128 and D = 1024 batch launches), which fetch
 * only the NEW neighbours' codes: [10]=queries (of [5]) handed to the re-run launch because the reference's stage-2
 * decision needed the rest of a block, [11]=expansions where that decision is unobservable and was left open -- there
 * [4] counts only the skips that were decided, a lower bound of the reference's counter (0 / exact for every other
 * instantiation). */
int cph_last_search_stats(cph_index* h, uint64_t out[12]);
/* Vertices expanded by each query of the last batch (its first pass); n = that batch's size. */
int cph_last_query_expansions(cph_index* index, uint32_t* out, uint64_t n);
/* Launch order of a batch (hook of the counting sort that hands queries out closest-entry-first):
 * order[n] = permutation of 0..n-1, ascending in the top 14 bits of the non-negative float keys. */
int cph_order_queries(cph_index* index, const float* keys, uint64_t n, uint32_t* order);

/* ---- kernel-level hooks ------------------------------------------------------------ */
/* Query encoder (encoder/rabitq_encoder.hpp:73-79,98-136,197-209): lut = u8[D/4][16] in
 * the reference's LUT format, coeffs = {coeff_fastscan, coeff_popcount, coeff_constant}. */
int cph_encode_query(cph_index* h, const float* query, uint8_t* lut, float* coeffs);

/* Upper-layer greedy descent (api/hnsw_index.hpp:196-202,617-638): layer-0 entry id. */
int cph_entry_point(cph_index* h, const float* query, uint32_t* entry);

/* One 32-neighbour FastScan block on the GPU for vertex `vertex` of the loaded index:
 * sums[32] (1-bit: plane sum; N-bit: weighted N-bit sum), msb[32] (plane-0 sum),
 * est[32]/lower[32] as search consumes them (stage-2 skip applied when nn_full != 0 and
 * no stage-1 lower bound is below `worst`: est = FLT_MAX, lower = stage-1 bound),
 * lower_stage1[32] (convert_msb_to_lower_bounds; == lower for 1-bit).
 * qparams = {coeff_fastscan, coeff_popcount, coeff_constant, affine_a, affine_b,
 * ip_qo_floor, dot_slack}.  (distance/fastscan_kernel.hpp:17-425,
 * search/rabitq_search.hpp:159-206) */
int cph_fastscan_block(cph_index* h, const uint8_t* lut, const float* qparams,
                       uint32_t vertex, float dist_qp_sq, float worst, int nn_full,
                       uint32_t* sums, uint32_t* msb, float* est, float* lower,
                       float* lower_stage1);

/* Exact L2 (search/rabitq_search.hpp:90-93, core/memory.hpp:81-95) of `query` against
 * nodes ids[0..n). */
int cph_exact_l2(cph_index* h, const float* query, const uint32_t* ids, uint64_t n,
                 float* out);

/* Streaming FastScan roofline benchmark on synthetic neighbour blocks (no graph).
 * create: allocates n_blocks device blocks of layout (D,bits) filled with seeded random
 * valid codes/aux.  run: `reps` passes over all blocks with one encoded query, both
 * N-bit stages per block; writes the average kernel time per pass (HIP events on the
 * launch stream), and a checksum.  block_bytes = device bytes per block. */
typedef struct cph_stream cph_stream;
int cph_fastscan_stream_create(int device, uint32_t D, uint32_t bits, uint64_t n_blocks,
                               uint64_t seed, cph_stream** out, uint64_t* block_bytes);
int cph_fastscan_stream_run(cph_stream* s, int reps, double* avg_ms, double* checksum);
/* Copies block `i` out in the REFERENCE neighbour-block layout
 * (distance/fastscan_layout.hpp:51-92,114-155) and the query (lut u8[D/4][16], 7 qparams,
 * dist_qp_sq) so a CPU implementation can be run on identical inputs. */
int cph_fastscan_stream_export(cph_stream* s, uint64_t first, uint64_t count,
                               uint8_t* ref_blocks, uint8_t* lut, float* qparams,
                               float* dist_qp_sq);
/* est/lower of block i as computed by the stream kernel (for parity checks). */
int cph_fastscan_stream_eval(cph_stream* s, uint64_t first, uint64_t count, float* est,
                             float* lower);
int cph_fastscan_stream_destroy(cph_stream* s);

/* ---- host-only hooks (no HIP call; used by the CPU test tier) --------------------------------- */
/* Reads a v2 index file with the library's reader and writes it back with its writer. */
int cph_host_rewrite_index(const char* path_in, const char* path_out);
/* Repacks one reference-layout neighbour block (distance/fastscan_layout.hpp:51-155) into the device
 * layout (dev_block, *dev_bytes bytes) and back into `ref_roundtrip`. */
int cph_host_repack_block(uint32_t D, uint32_t bits, const uint8_t* ref_block, uint8_t* dev_block,
                          uint64_t* dev_bytes, uint8_t* ref_roundtrip);
/* Host mirror of the query encoder (the device encoder is cph_encode_query): lut u8[D/4][16],
 * coeffs[3], masks u32[max(1,D/32)][4] (the bit-sliced form the kernels consume). */
int cph_host_encode_query(uint64_t dim, const float* query, uint8_t* lut, float* coeffs, uint32_t* masks);

#ifdef __cplusplus
}
#endif
#endif /* CPHNSW_MI355X_H */
