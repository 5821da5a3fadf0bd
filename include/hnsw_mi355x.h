/* hnsw_mi355x.h -- C ABI of libhnsw_mi355x.so
 *
 * MI355X-native drop-in for ONE hot path of the Rust crate hnsw_rs 0.3.4
 * (jean-pierreBoth/hnswlib-rs): the batched k-NN search
 *   AnnT::parallel_search_neighbours          src/api.rs:58-65
 *   -> Hnsw::parallel_search                  src/hnsw.rs:1612-1635
 *   -> Hnsw::search_filter(filter = None)     src/hnsw.rs:1487-1580
 *   -> Hnsw::search_layer                     src/hnsw.rs:922-1064
 *   -> Distance<f32>::eval (DistL2/DistCosine/DistDot/DistL1, crate anndists 0.1)
 * together with the hnswio dump format on either side of it (src/hnswio.rs).
 *
 * Two groups of entry points:
 *   (1) hnswgpu_*  : the thin ABI a Rust `impl AnnT` wrapper (or any host) binds; flat
 *                    row-major matrices in, flat arrays out, integer status codes.
 *   (2) the reference's own f32 C symbols (src/libext.rs), name- and layout-compatible,
 *                    implemented on top of (1) -- existing C / Julia callers relink as is.
 *
 * All pointers are plain host pointers unless the name says `_device`.  No function aborts
 * the process (the reference calls process::exit / panics): failures
 * return a status / NULL and hnswgpu_last_error() explains.
 * The search entry points REQUIRE a gfx950 device: there is no CPU fallback.
 */
#ifndef HNSW_MI355X_H
#define HNSW_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes ------- */
enum {
    HNSWGPU_OK = 0,
    HNSWGPU_ERR_ARG = 1,      /* bad argument (null, dimension mismatch, ...)             */
    HNSWGPU_ERR_IO = 2,       /* cannot open / read / write a dump file                   */
    HNSWGPU_ERR_FORMAT = 3,   /* bad magic, truncated file, incoherent ids                */
    HNSWGPU_ERR_DISTANCE = 4, /* dump's distance name differs from the one asked          */
    HNSWGPU_ERR_TYPE = 5,     /* dump's element type is not "f32"                         */
    HNSWGPU_ERR_DEVICE = 6,   /* HIP error, or no gfx950 device / index not uploaded      */
    HNSWGPU_ERR_EMPTY = 7,    /* operation needs a non-empty index                        */
    HNSWGPU_ERR_REF_PANIC = 8 /* filtered search: the reference panics on some of the queries
                                 (src/hnsw.rs:973) and no per-query status array was given  */
};

/* metric selector == short type name of the anndists distance */
enum {
    HNSWGPU_DIST_L2 = 0,     /* anndists::dist::distances::DistL2     sqrt(sum (a-b)^2)   */
    HNSWGPU_DIST_COSINE = 1, /* ...::DistCosine   1 - a.b/sqrt(|a|^2 |b|^2), f64 sums     */
    HNSWGPU_DIST_DOT = 2,    /* ...::DistDot      1 - a.b (inputs pre-normalised)         */
    HNSWGPU_DIST_L1 = 3,     /* ...::DistL1       sum |a-b|                               */
    /* distances between probability vectors (the f32 arms of src/libext.rs:334-345, :491-513)        */
    HNSWGPU_DIST_HELLINGER = 4,     /* ...::DistHellinger      sqrt(max(1 - sum sqrt(a) sqrt(b), 0))          */
    HNSWGPU_DIST_JEFFREYS = 5,      /* ...::DistJeffreys       sum (a-b) ln(max(a,1e-30)/max(b,1e-30))        */
    HNSWGPU_DIST_JENSENSHANNON = 6  /* ...::DistJensenShannon  sqrt(0.5 sum [a ln(a/m) + b ln(b/m)]), m=(a+b)/2 */
};

typedef struct hnswgpu_index hnswgpu_index; /* opaque: flat host graph + its HBM replica  */

/* Thread-local message for the last failing call on this thread. */
const char* hnswgpu_last_error(void);

/* ---------------------------------------------------------------- load / dump -------- */
/* HnswIo::new(dir, basename) + load_hnsw::<f32, D>()        src/hnswio.rs:317, :431-524
 * `dist` = HNSWGPU_DIST_* the caller asks for (checked against the dump's distname by the
 * short-name rule of src/hnswio.rs:473-490), or -1 to accept whatever the dump says.     */
int hnswgpu_load_dump(const char* dir, const char* basename, int dist, hnswgpu_index** out);

/* AnnT::file_dump(path, basename) in DumpMode::Full           src/api.rs:70-93,
 * src/hnswio.rs:1355-1387.  Overwrites existing files (no unique-name logic).            */
int hnswgpu_file_dump(const hnswgpu_index* idx, const char* dir, const char* basename);

void hnswgpu_free_index(hnswgpu_index* idx);

/* Description of a dump (load_description, src/hnswio.rs:937-1042).                       */
typedef struct {
    uint32_t format_version; /* 2, 3 or 4 (from the magic)                                */
    uint8_t dumpmode;        /* 1 = Full                                                  */
    uint8_t max_nb_connection;
    uint8_t nb_layer;
    double level_scale;
    uint64_t ef_construction;
    uint64_t nb_point;
    uint64_t dimension;
    char distname[260];
    char t_name[260];
} hnswgpu_description;
int hnswgpu_load_description(const char* graph_file_path, hnswgpu_description* out);
int hnswgpu_get_description(const hnswgpu_index* idx, hnswgpu_description* out);

/* ---------------------------------------------------------------- DataMap ------------ */
/* DataMap (src/datamap.rs:24-319): the vectors of a dump memory-mapped and addressed by DataId, without loading the
 * graph.  open = DataMap::from_hnswdump::<f32>(dir, basename) (:44-231; dumps of format > 2 whose type is f32; where
 * the reference exits the process on a missing file this returns HNSWGPU_ERR_IO); get_data = get_data::<f32>(&id)
 * (:276-297): a pointer to `dimension` floats inside the mapping, valid until close, NULL for an unknown id;
 * ids = get_dataid_iter (:301): the ids in file order (returns their number; fills at most cap).                    */
typedef struct hnswgpu_datamap hnswgpu_datamap;
int hnswgpu_datamap_open(const char* dir, const char* basename, hnswgpu_datamap** out);
void hnswgpu_datamap_close(hnswgpu_datamap* m);
const float* hnswgpu_datamap_get_data(const hnswgpu_datamap* m, uint64_t data_id);
uint64_t hnswgpu_datamap_nb_data(const hnswgpu_datamap* m);
uint64_t hnswgpu_datamap_dimension(const hnswgpu_datamap* m);
const char* hnswgpu_datamap_distname(const hnswgpu_datamap* m);
const char* hnswgpu_datamap_typename(const hnswgpu_datamap* m);
uint64_t hnswgpu_datamap_ids(const hnswgpu_datamap* m, uint64_t* out, uint64_t cap);

/* ---------------------------------------------------------------- construction ------- */
/* Hnsw::<f32, D>::new(max_nb_connection, max_elements, max_layer, ef_construction, D)
 * + (parallel_)insert of n points (src/hnsw.rs:771, :1077-1215, :1224-1238).  Construction on
 * the host cores, or GPU-assisted (gpu_device); levels come from the documented
 * SplitMix64(397) stream.                                                */
typedef struct {
    uint64_t max_nb_connection; /* M; layer-0 lists hold up to 2M                         */
    uint64_t ef_construction;
    uint64_t max_layer;         /* clamped to 16; must be 16 for the index to be dumpable */
    int dist;                   /* HNSWGPU_DIST_*                                         */
    double level_scale_factor;  /* modify_level_scale(), in [0.2, 1]; 1.0 = default       */
    int extend_candidates;      /* set_extend_candidates()                                */
    int keep_pruned;            /* set_keeping_pruned()                                   */
    int nthreads;               /* 1 = serial insert (deterministic); 0 = all host cores  */
    int fast_arithmetic;        /* 0: reference-order scalar sums; 1: 8-lane SIMD sums
                                   (the crate's `simdeez_f` build order)                  */
    int gpu_assist;             /* 1: GPU-assisted construction: the searches of every insertion
                                   (src/hnsw.rs:1114-1197) run on HIP device gpu_device, window by window against a
                                   frozen snapshot of the graph; select_neighbours, list and reverse updates on the
                                   host cores.  0 (default): host only.  Device distances are reference-order.      */
    int gpu_device;
    uint64_t gpu_window;        /* points per window at most (0 = 16384); windows grow with the graph
                                   (max(256, inserted / 8)); 1 = one point at a time: the serial insertion, exactly */
} hnswgpu_build_params;
int hnswgpu_build(const float* data, uint64_t n, uint64_t d, const uint64_t* ids /* NULL: 0..n-1 */,
                  const hnswgpu_build_params* params, hnswgpu_index** out);
/* Hnsw::insert / parallel_insert of n more points into ANY index (src/hnsw.rs:1068-1075, :1224-1238), including one
 * obtained from hnswgpu_load_dump: like the reference's reloaded Hnsw it keeps growing (M, ef_construction and the
 * level scale come from the dump; extend_candidates = true, keep_pruned = false as src/hnswio.rs:510-511 sets them;
 * the reference's reload quirk of treating the dumped absolute level scale as a factor, src/hnswio.rs:773-777, is
 * kept).  nthreads: 1 = serial, 0 = all host cores.  HBM replicas are refreshed by the next search / upload.           */
int hnswgpu_insert(hnswgpu_index* idx, const float* data, uint64_t n, uint64_t d, const uint64_t* ids /* NULL: continue */,
                   int nthreads);
/* the same, GPU-assisted (see hnswgpu_build_params.gpu_assist / gpu_device / gpu_window).  What can be refused up front (no
 * device, bad ordinal, ef_construction > 1024, wrong dimension) is refused before a point is accepted: the index is then
 * unchanged and a retry is safe.  A device failure AFTER the points were accepted (allocation, copy, kernel) does not lose
 * them: the host builder links the rest of the batch, the call returns HNSWGPU_OK and hnswgpu_last_error() holds the device's
 * message as a warning.  Either way hnswgpu_nb_point() counts fully linked points only.                                    */
int hnswgpu_insert_gpu(hnswgpu_index* idx, const float* data, uint64_t n, uint64_t d, const uint64_t* ids, int nthreads,
                       int gpu_device, uint64_t gpu_window);

/* ---------------------------------------------------------------- inspection --------- */
uint64_t hnswgpu_nb_point(const hnswgpu_index* idx);
uint64_t hnswgpu_dimension(const hnswgpu_index* idx);
int hnswgpu_dist(const hnswgpu_index* idx);
uint64_t hnswgpu_layer_nb_point(const hnswgpu_index* idx, unsigned layer);
int hnswgpu_max_level_observed(const hnswgpu_index* idx);
/* entry point as (origin_id, layer, rank); returns HNSWGPU_ERR_EMPTY on an empty index    */
int hnswgpu_entry_point(const hnswgpu_index* idx, uint64_t* origin_id, uint8_t* layer, int32_t* rank);
/* neighbour list of point (layer, rank) at layer `l`: fills up to cap entries, returns the
 * list length (or -1).  Order = stored order (ascending stored distance).                 */
int64_t hnswgpu_neighbours(const hnswgpu_index* idx, unsigned layer, int32_t rank, unsigned l, uint64_t cap,
                           uint64_t* origin_ids, uint8_t* layers, int32_t* ranks, float* dists);

/* ---------------------------------------------------------------- HBM replica -------- */
/* Copies vectors (rows padded to 128-byte lines) and neighbour lists into the HBM of HIP
 * device `device`.  Idempotent.  A handle may hold replicas on several devices (one upload
 * each); the last device uploaded explicitly is the PRIMARY one, used by the single-device
 * search entry points.  Replicas are immutable; they are dropped when the graph changes.    */
int hnswgpu_upload(hnswgpu_index* idx, int device);
int hnswgpu_device_count(void);

/* ---------------------------------------------------------------- search ------------- */
/* Hnsw::parallel_search(datas, knbn, ef) for a flat nq x d row-major query matrix
 * (src/hnsw.rs:1612-1635).  Row i of the outputs holds out_counts[i] <= k valid entries in
 * ascending distance: Neighbour{d_id, distance, p_id(layer, rank)} (src/hnsw.rs:98-107).
 * out_layer / out_rank may be NULL.  Host buffers; includes H2D/D2H copies.               */
int hnswgpu_search_batch(const hnswgpu_index* idx, const float* queries, uint64_t nq, uint64_t d, uint64_t k,
                         uint64_t ef, uint64_t* out_ids, float* out_dists, uint8_t* out_layer, int32_t* out_rank,
                         uint32_t* out_counts);

/* Hnsw::search_filter(data, knbn, ef, Some(&Vec<usize>)) for every query of a batch (src/hnsw.rs:1487-1580; the filter
 * is `impl FilterT for Vec<usize>`, a binary search in a SORTED id vector, src/filter.rs:11-15 -- an unsorted vector is
 * HNSWGPU_ERR_ARG).  Same outputs as hnswgpu_search_batch.  The reference panics on some inputs (a filter that empties
 * return_points, then `peek().unwrap()`, src/hnsw.rs:973 -- only reachable with ef == 1): such a query gets count 0
 * and out_status[i] = 1 (0 otherwise); with out_status == NULL the call returns HNSWGPU_ERR_REF_PANIC instead (the other
 * queries' answers are valid).  Never aborts.                                                                      */
int hnswgpu_search_batch_filtered(const hnswgpu_index* idx, const float* queries, uint64_t nq, uint64_t d, uint64_t k,
                                  uint64_t ef, const uint64_t* allowed_ids, uint64_t n_allowed, uint64_t* out_ids,
                                  float* out_dists, uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts,
                                  uint8_t* out_status);

/* Hnsw::parallel_search with the batch sharded over several GPUs of THIS process (BASELINE config 4 without Python or
 * a collective library): the graph is replicated on every device named in devices[0..n_shards) (uploaded on first use),
 * shard s = the s-th contiguous balanced block of queries (nq / n_shards each, the first nq % n_shards one more), one
 * host thread per shard, and every shard copies its answers straight into its rows of the caller's arrays -- that is
 * the gather.  A device may be named more than once (shards then share its replica and run concurrently).
 * Answers are identical to hnswgpu_search_batch on one device.                                                      */
int hnswgpu_search_batch_sharded(const hnswgpu_index* idx, const int* devices, int n_shards, const float* queries,
                                 uint64_t nq, uint64_t d, uint64_t k, uint64_t ef, uint64_t* out_ids, float* out_dists,
                                 uint8_t* out_layer, int32_t* out_rank, uint32_t* out_counts);
/* The same for a host that already holds its data in HBM: shard s searches nq_shard[s] queries that sit at d_queries[s] ON
 * devices[s] and leaves its answers in that device's d_out_*[s] arrays (nq_shard[s] x k; d_out_layer / d_out_rank and their
 * entries may be NULL) -- no PCIe traffic but the counters.  One host thread per shard; streams[s] (hipStream_t as void*,
 * may be NULL, the array too) is the stream shard s launches on.  Missing replicas are uploaded first, all devices at once.
 * What is exchanged afterwards (answers only: nq x k x 12 bytes) is the caller's collective, e.g. RCCL all-gather.         */
int hnswgpu_search_batch_sharded_device(const hnswgpu_index* idx, const int* devices, int n_shards, const float* const* d_queries,
                                        const uint64_t* nq_shard, uint64_t d, uint64_t k, uint64_t ef, uint64_t* const* d_out_ids,
                                        float* const* d_out_dists, uint8_t* const* d_out_layer, int32_t* const* d_out_rank,
                                        uint32_t* const* d_out_counts, void* const* streams);
/* The exchange that follows hnswgpu_search_batch_sharded_device when ONE process drives the node's GPUs and holds no collective
 * library (a Rust host): shard s's answers (nq_shard[s] x k on devices[s]) are copied behind one another -- input order, the order
 * Hnsw::parallel_search returns (src/hnsw.rs:1623-1633) -- into the arrays of root_device with hipMemcpyPeerAsync on root_stream
 * (xGMI between the GPUs of a node), and the stream is waited for.  1.2-1.7 MB per shard at BASELINE config 4.  d_layer / d_rank,
 * their entries and root_layer / root_rank may be NULL.                                                                       */
int hnswgpu_gather_sharded_answers(const int* devices, int n_shards, const uint64_t* nq_shard, uint64_t k,
                                   const uint64_t* const* d_ids, const float* const* d_dists, const uint8_t* const* d_layer,
                                   const int32_t* const* d_rank, const uint32_t* const* d_counts, int root_device,
                                   uint64_t* root_ids, float* root_dists, uint8_t* root_layer, int32_t* root_rank,
                                   uint32_t* root_counts, void* root_stream);

/* Same with every buffer already resident in HBM (device pointers), launched on HIP stream
 * `stream` (hipStream_t as void*; NULL = default stream).  Synchronises `stream` once
 * before returning (the visited-set overflow check needs one 4-byte read-back).
 * d_stats may be NULL, else uint32[nq*8] per query = {n_dist, n_expand, n_ids_read, status,
 * t_start, t_end (device wall clock, 10 ns ticks), used_hbm_bitmap, flags | (lists scanned by the greedy descent << 8) | (its n_dist << 16)}
 * (flags: 1 equal distances met, 2 a pop was taken from the literal candidate heap; n_dist / n_expand / n_ids_read include
 * the descent, which runs in a kernel of its own in front of the search kernel; its ids read = its n_dist - 1).
 * status: 0 ok; 2 ok, but the answer depends on the reference's heap order and strict ties are off;
 * 3 ok, resolved with the literal heaps (strict ties; see DESIGN.md "ties").  ef above 1024 (the
 * register-resident result set) is served by the literal-heap kernel: correct, slower.          */
int hnswgpu_search_batch_device(const hnswgpu_index* idx, const float* d_queries, uint64_t nq, uint64_t d,
                                uint64_t k, uint64_t ef, uint64_t* d_out_ids, float* d_out_dists,
                                uint8_t* d_out_layer, int32_t* d_out_rank, uint32_t* d_out_counts,
                                uint32_t* d_stats, void* stream);
/* hnswgpu_search_batch_filtered on device-resident buffers (the sorted id vector too).  d_stats[q*8+3] == 6 marks a
 * query on which the reference panics; *n_panics (may be NULL) counts them.                                        */
int hnswgpu_search_batch_filtered_device(const hnswgpu_index* idx, const float* d_queries, uint64_t nq, uint64_t d,
                                         uint64_t k, uint64_t ef, const uint64_t* d_allowed_ids, uint64_t n_allowed,
                                         uint64_t* d_out_ids, float* d_out_dists, uint8_t* d_out_layer,
                                         int32_t* d_out_rank, uint32_t* d_out_counts, uint32_t* d_stats, void* stream,
                                         uint32_t* n_panics);
/* The same call, not waited for: hnswgpu_search_batch_device_begin returns at once with a ticket (a worker thread of
 * the library issues the launches on `stream` and waits for them), hnswgpu_search_batch_end waits for that call, returns
 * its status (its message is then hnswgpu_last_error() of the caller's thread) and releases the ticket.  Two batches in
 * flight from one host thread keep the device full while a launch drains towards its longest search (DESIGN.md
 * section 8).  Every ticket must be ended exactly once; the buffers and the index must stay alive until then; give
 * concurrent tickets different output buffers (and, to let them overlap on the device, different streams).          */
typedef struct hnswgpu_ticket hnswgpu_ticket;
int hnswgpu_search_batch_device_begin(const hnswgpu_index* idx, const float* d_queries, uint64_t nq, uint64_t d,
                                      uint64_t k, uint64_t ef, uint64_t* d_out_ids, float* d_out_dists,
                                      uint8_t* d_out_layer, int32_t* d_out_rank, uint32_t* d_out_counts,
                                      uint32_t* d_stats, void* stream, hnswgpu_ticket** ticket);
int hnswgpu_search_batch_end(hnswgpu_ticket* ticket);
/* Concurrency: the search entry points may be called from several host threads on ONE handle at the same time (the
 * reference's search is `&self`); every call takes a private workspace.  Calls that change the index (insert, upload)
 * wait for running searches.                                                                                       */

/* Ties.  Two EQUAL f32 distances can make the reference's answer depend on the internal order of Rust's BinaryHeap.
 * The kernel follows the reference by VALUE and detects the places where values do not decide (two equal nearest
 * candidates at a pop, an equal candidate evicted from the result set that is still the farthest distance, equal
 * distances or an ambiguous survivor among the first k answers).  With strict ties ON (default; env
 * HNSWGPU_STRICT_TIES=0 or this call turns it off) such a query carries on, inside the same launch, with a literal
 * emulation of the heap in question rebuilt from a log of the search so far (stats status 3), so that ids match the
 * reference bit for bit also under ties.  OFF: equals are ordered by arrival and the query is flagged (status 2).
 * hnswgpu_last_tie_count: queries of the last call that met such a place (resolved, or flagged).                   */
int hnswgpu_set_strict_ties(hnswgpu_index* idx, int on);
int hnswgpu_last_tie_count(const hnswgpu_index* idx, uint32_t* ties);

/* The HNSWGPU_* tuning and test hooks (HNSWGPU_HASH_BITS, _NO_SCHED, _NO_INKERNEL, _STRICT_WG_PER_CU, _CAND_LDS, _WAVES_PER_CU,
 * _EXACT_FIRST, _TRACE_LAUNCH, _TRACE_HOST, _HOST_THREADS, _HOST_CHUNKS, _FFI_UNPACK) are read from the environment ONCE per
 * process, at the library's first search -- never on the launch path.  A caller that changes them afterwards (the tests do)
 * says so with this call.  Always HNSWGPU_OK.                                                                          */
int hnswgpu_reload_env(void);

/* The arithmetic of Distance<f32>::eval during SEARCH (construction always sums like the scalar build).
 *   HNSWGPU_ARITH_SCALAR (default): the crate's default build -- every sum left to right over the vector index
 *                                   (anndists 0.1 without features, Cargo.toml:104-106); what the parity tests pin.
 *   HNSWGPU_ARITH_SIMD8:            an APPROXIMATION of the summation order of its `simdeez_f` / `stdsimd` builds
 *                                   (Cargo.toml:107-111; the builds behind every number the reference publishes,
 *                                   README.md:46-56): 8 vertical f32 accumulators over the full blocks of 8 elements, their
 *                                   horizontal sum LEFT TO RIGHT, then the d % 8 tail; DistCosine with three such f32 sums.
 *                                   The crate's sources are not available here: its vector width follows the ISA chosen at
 *                                   run time, its horizontal add may fold halves pairwise, and DistCosine may have no SIMD
 *                                   kernel at all -- so this mode is checked against the oracle's dist_simd8 only, and is
 *                                   NOT claimed to reproduce a simdeez_f build bit for bit until oracle/ref_pin has been run
 *                                   with `--features simdeez_f` (.github/workflows/pin.yml does).  DistL2 / DistCosine / DistDot / DistL1 only
 *                                   (other distances keep the scalar order).  Last-bit differences to the scalar build, so
 *                                   near-tie ids may differ from it; the checker for this mode is the oracle's dist_simd8.
 * Opt-in, never the default.  Takes effect for the following search calls on every replica of the index.            */
#define HNSWGPU_ARITH_SCALAR 0
#define HNSWGPU_ARITH_SIMD8 1
int hnswgpu_set_arithmetic(hnswgpu_index* idx, int arithmetic);

/* Timing of the kernels of the last search call on this index, measured with HIP events on
 * the launch stream: total milliseconds and number of launches (retries included).        */
int hnswgpu_last_kernel_ms(const hnswgpu_index* idx, double* ms, uint32_t* launches);
/* Same, but only up to the end of the search kernel proper (excludes the exact replay of tied queries). */
int hnswgpu_last_search_kernel_ms(const hnswgpu_index* idx, double* ms);

/* Distance<f32>::eval evaluated ON THE DEVICE by the search kernel's own distance routine (batch_dist: lane groups of
 * 4 or 2 lanes per row, left-to-right sum in the group's first lane) -- for arithmetic parity tests (host buffers).
 *   hnswgpu_eval_distances:       out[i]    = dist(a[i], b[i]), n pairs (each one a batch of a single row)
 *   hnswgpu_eval_distance_matrix: out[q][r] = dist(queries[q], rows[r]); the rows are evaluated in batches of `batch`
 *                                 (1..64) consecutive rows, the branch structure the search takes for that many fresh
 *                                 neighbours (<= 16 rows: 4 lanes per row; more: 32 rows at 2 lanes per row, ...).  */
int hnswgpu_eval_distances(int dist, const float* a, const float* b, uint64_t n, uint64_t d, float* out);
int hnswgpu_eval_distance_matrix(int dist, const float* queries, uint64_t nq, const float* rows, uint64_t n, uint64_t d,
                                 uint32_t batch, float* out);
/* the same in the given arithmetic (HNSWGPU_ARITH_*)                                                                  */
int hnswgpu_eval_distance_matrix_arith(int dist, int arithmetic, const float* queries, uint64_t nq, const float* rows,
                                       uint64_t n, uint64_t d, uint32_t batch, float* out);

/* Test entry: the lane lab.  The wave-level algorithms the search kernels are made of -- the reference's BinaryHeap (src/hnsw.rs:940,
 * :958-973, :1035-1053, :1544; std's push / pop / into_sorted_vec) as a memory heap and as a register heap, the sorted result set
 * with its accept rule (:1028-1053), the exact visited table (:955-956, :1016-1017) -- run on `device` by ONE wavefront from a
 * script: ops[n_ops][4] = {op, a, b, c}, lanes[n_lane_sets][64][2] = {f32 bits, id} (the lane vectors of the batch operations).
 * mode 0 memory heap (p0 entries in LDS, p1 pop variant), 1 register heap (p0 slots per lane), 2 result set (p0 slots per lane,
 * p1 ef), 3 visited table (p0 tbits, p1 idbits, p2 restbits).  out[0] = words produced, then every operation's result in script
 * order, then the final state.  Operation codes and layouts: hnswlib-rs_amd/csrc/search_kernels.hpp (LAB_*), lane_lab.inc;
 * tests/test_gpu_lane_lab.py drives it.                                                                                    */
int hnswgpu_lane_lab(int device, uint32_t mode, uint32_t p0, uint32_t p1, uint32_t p2, const uint32_t* ops, uint32_t n_ops,
                     const uint32_t* lanes, uint32_t n_lane_sets, uint32_t* out, uint32_t out_words);


/* =======================================================================================
 * (2) The reference's own C ABI for f32 (src/libext.rs), same names and struct layouts.
 * ======================================================================================= */
typedef struct HnswIo HnswIo;         /* src/hnswio.rs:300-310 (opaque)                   */
typedef struct HnswApif32 HnswApif32; /* src/libext.rs:106 (opaque)                        */

typedef struct { /* src/libext.rs:64-71 */
    size_t id;
    float d;
} Neighbour_api;
typedef struct { /* src/libext.rs:82-87 */
    int64_t nbgh;
    const Neighbour_api* neighbours;
} Neighbourhood_api;
typedef struct { /* Vec_api<Neighbourhood_api>, src/libext.rs:58-62 */
    int64_t len;
    const Neighbourhood_api* ptr;
} Vec_api_Neighbourhood;
typedef struct { /* src/libext.rs:1121-1141 */
    uint8_t dumpmode;
    uint8_t max_nb_connection;
    uint8_t nb_layer;
    size_t ef;
    size_t nb_point;
    size_t data_dimension;
    size_t distname_len;
    const uint8_t* distname;
    size_t t_name_len;
    const uint8_t* t_name;
} DescriptionFFI;

const HnswIo* get_hnswio(uint64_t flen, const uint8_t* name);                      /* :28-33   */
const HnswApif32* load_hnswdump_f32_DistL1(HnswIo* hnswio);                        /* :310-315 */
const HnswApif32* load_hnswdump_f32_DistL2(HnswIo* hnswio);                        /* :316-321 */
const HnswApif32* load_hnswdump_f32_DistCosine(HnswIo* hnswio);                    /* :322-327 */
const HnswApif32* load_hnswdump_f32_DistDot(HnswIo* hnswio);                       /* :328-333 */
const HnswApif32* load_hnswdump_f32_DistJensenShannon(HnswIo* hnswio);             /* :334-339 */
const HnswApif32* load_hnswdump_f32_DistJeffreys(HnswIo* hnswio);                  /* :340-345 */
const HnswApif32* init_hnsw_f32(size_t max_nb_conn, size_t ef_const, size_t namelen,
                                const uint8_t* cdistname);                         /* :458-523 */
const HnswApif32* new_hnsw_f32(size_t max_nb_conn, size_t ef_const, size_t namelen, const uint8_t* cdistname,
                               size_t max_elements, size_t max_layer);             /* :532-580 */
/* The crate's init_hnsw_ptrdist_f32 takes a HOST function pointer as the distance (DistCFFI<f32>).  A host callback
 * cannot be evaluated by the device, and this library has no CPU search path by design: the symbol exists so that a host
 * linking the crate's C interface resolves, it always returns NULL and sets HNSWGPU_ERR_DISTANCE (hnswgpu_last_error()
 * says why).  Use one of the named distances. */
typedef float (*hnsw_dist_fn_f32)(const float* a, const float* b, unsigned long long len);
const HnswApif32* init_hnsw_ptrdist_f32(size_t max_nb_conn, size_t ef_const, hnsw_dist_fn_f32 c_func);  /* :643-655 */
void insert_f32(HnswApif32* hnsw_api, size_t len, const float* data, size_t id);   /* :661-678 */
void parallel_insert_f32(HnswApif32* hnsw_api, size_t nb_vec, size_t vec_len, const float** datas,
                         const size_t* ids);                                       /* :683-723 */
const Neighbourhood_api* search_neighbours_f32(const HnswApif32* hnsw_api, size_t len, const float* data,
                                               size_t knbn, size_t ef_search);     /* :728-767 */
const Vec_api_Neighbourhood* parallel_search_neighbours_f32(const HnswApif32* hnsw_api, size_t nb_vec,
                                                            int64_t vec_len, const float** data, size_t knbn,
                                                            size_t ef_search);     /* :205-254 */
int64_t file_dump_f32(const HnswApif32* hnsw_api, size_t namelen, const uint8_t* filename); /* :257-275 */
void drop_hnsw_f32(const HnswApif32* p);                                           /* :626-630 */
const DescriptionFFI* load_hnsw_description(size_t flen, const uint8_t* name);     /* :1171-1232 */
void init_rust_log(void);                                                          /* :1238-1240 (no-op) */

/* The reference leaks every result buffer to the caller and exports no free function
 * (SURVEY.md 8b "Ownership").  These are additions.
 * hnswgpu_free_neighbourhood releases what search_neighbours_f32 returned (the struct and its row).
 * hnswgpu_free_neighbourhood_vec releases what parallel_search_neighbours_f32 returned, and is the ONLY way to release
 * it: the answer is one allocation (Vec_api | Neighbourhood_api[nb_vec] | every Neighbour_api row) -- usually page-locked
 * memory the search kernels wrote in place, which free() cannot release at all --, so `ptr` and each
 * `neighbours` are interior pointers -- never free() a row or hand one of its Neighbourhood_api to
 * hnswgpu_free_neighbourhood, and never pass a Vec_api this library did not return.                                   */
void hnswgpu_free_neighbourhood(const Neighbourhood_api* p);
void hnswgpu_free_neighbourhood_vec(const Vec_api_Neighbourhood* p);
void hnswgpu_free_hnswio(const HnswIo* p);
void hnswgpu_free_description(const DescriptionFFI* p);
/* access the thin-ABI handle behind a reference-style handle (NULL until built/loaded)    */
hnswgpu_index* hnswgpu_from_api(const HnswApif32* p);

#ifdef __cplusplus
}
#endif
#endif /* HNSW_MI355X_H */
