/*
 * ipkgpu.h -- C ABI of the MI355X phylo-k-mer scoring engine (libipkgpu.so).
 *
 * Drop-in boundary for ONE path of phylo42/IPK: what db_builder::explore_kmers /
 * explore_group consume from the scoring layer (citations relative to the IPK tree):
 *
 *   ipk/src/db_builder.cpp:576-627   explore_kmers   (the group loop)
 *   ipk/src/db_builder.cpp:629-698   explore_group   (windows -> DCLA -> put -> DB insert)
 *   ipk/src/pk_compute.cpp:28-119    DCLA::run / DC  (divide-and-conquer with look-ahead)
 *   ipk/src/window.cpp:16-27,69-72   matrix::preprocess / range_max_sum
 *   ipk/src/branch_group.cpp:88-107  put / kmer_batch
 *
 * IPK itself has no FFI; INTEGRATION.md shows the ~30-line patch to db_builder.cpp that
 * binds these entry points.  Plain pointers and sizes only; no exceptions cross the ABI:
 * every call returns an int status (0 = ok) and ipkgpu_last_error() gives the message.
 *
 * Threading: one host thread per context; calls on a context are serialised by the caller.
 * One context drives one GPU (one process per GPU; multi-GPU sharding is by branch group).
 *
 * Streams: a context works on its OWN non-blocking HIP stream.  Device buffers handed in by raw
 * pointer (logp_dev, counts_dev, entries_dev ...) must be complete before the call: synchronise the
 * stream that produced them (hipStreamSynchronize / hipEventSynchronize) first.  Every call returns
 * after its device work has finished, so results may be read from any stream afterwards.
 *
 * Not supported (IPKGPU_ERR_INVALID, never a silent fallback): DNA k > 14, amino acids k > 6 (the reference's
 * command line advertises k <= 31, ipk.py:116; its key type here is u32: DNA k <= 16) -- at DNA k = 13, 14 also a
 * call in which one window's half list (its 6- / 7-symbol prefixes or suffixes above their threshold) exceeds 6144
 * entries: near-uniform columns, which real posteriors do not have -- and the reference's on-disk mode (db_builder.cpp:673-681, branch_group.cpp:109-185: per-group
 * files merged later): groups are batched by device memory instead ("workspace_bytes") and the k-mer-keyed
 * merge of batches / ranks runs on the device (ipkgpu_merge_parts*).
 */
#ifndef IPKGPU_H
#define IPKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ipkgpu_ctx ipkgpu_ctx;
typedef struct ipkgpu_result ipkgpu_result;
typedef struct ipkgpu_parts ipkgpu_parts;
typedef struct ipkgpu_db ipkgpu_db;

enum {
    IPKGPU_OK = 0,
    IPKGPU_ERR_INVALID = 1,     /* bad argument (sigma/k unsupported, sites < k, null pointer ...) */
    IPKGPU_ERR_HIP = 2,         /* a HIP runtime call failed */
    IPKGPU_ERR_NOMEM = 3,       /* device or host allocation failed */
    IPKGPU_ERR_NODEVICE = 4     /* no usable GPU: the engine has NO CPU fallback */
};

/* ipkgpu_result_time_ms selectors */
enum {
    IPKGPU_T_TOTAL = 0,         /* whole call on the device stream (first event -> last event) */
    IPKGPU_T_PREFIX = 1,        /* prefix-of-column-maxima kernel  (matrix::preprocess) */
    IPKGPU_T_SCORE = 2,         /* scoring + max-reduce kernels (DCLA + put), summed over batches */
    IPKGPU_T_COMPACT = 3,       /* table -> sorted (key, score) compaction kernels */
    IPKGPU_T_SCORE_LAUNCHES = 4,/* number of scoring passes (batches of groups) the sums cover */
    IPKGPU_T_SCORE_MAIN = 5,    /* the dominant kernel alone: list building + pair emission (score_stream_kernel /
                                   score_tiles_kernel), summed over batches */
    IPKGPU_T_SCORE_REDUCE = 6,  /* per-bucket LDS max-reduce (reduce_buckets_kernel / reduce_ranges_kernel) */
    IPKGPU_T_XP_COUNT = 7,      /* exact-partition variant: the count pass alone (parts only) */
    IPKGPU_T_XP_WRITE = 8,      /* exact-partition variant: the write pass alone (parts only) */
    IPKGPU_T_KM_WRITE = 9       /* key-major writer kernel alone (km_write_kernel / km_write_c_kernel; parts only) */
};

/* ---- context ------------------------------------------------------------------------- */

/* Creates an engine on HIP device `device_id`.  Fails with IPKGPU_ERR_NODEVICE when no GPU
 * is present.  (Replaces nothing in IPK; owns the stream and workspaces.) */
int ipkgpu_create(int device_id, ipkgpu_ctx** ctx);
void ipkgpu_destroy(ipkgpu_ctx* ctx);

/* Message of the last failing call on this context ("" if none); owned by the context.
 * ipkgpu_last_error(NULL) returns the message of the last failed ipkgpu_create. */
const char* ipkgpu_last_error(const ipkgpu_ctx* ctx);
/* Name of the dominant kernel of the context's last scoring call ("score_quad_kernel", "score_stream_kernel",
 * "score_xp_kernel", "score_tiles_kernel"): what IPKGPU_T_SCORE_MAIN timed. */
const char* ipkgpu_last_main_kernel(const ipkgpu_ctx* ctx);
/* 1 if the last scoring batch left its per-group tables in the compressed form (occupancy bits + rank + score codes: AA k=6,
 * DNA k=11, 12), 0 for dense tables: tells the benchmark which reduce / key-major writer kernels ran. */
int ipkgpu_last_tables_compressed(const ipkgpu_ctx* ctx);

/* Diagnostics: -1 in the shipped build; in an IPK_EXEC_ASSERT build (ipk_amd/build.py, variant "execassert") the number of
 * times one of the kernels' exec-writing inline-asm helpers was entered with a partial exec mask (must be 0). */
int64_t ipkgpu_debug_exec_violations(ipkgpu_ctx* ctx);

/* Options: "workspace_bytes" (max bytes of per-group score tables resident at once; groups are
 * processed in batches that fit); "variant" (0 = auto: LDS max-reduce fed by the chunked pair pool, or by
 * the exact-partition passes for AA k=6; 1 = global-atomic max-reduce; 2 = force the chunked pool;
 * 3 = force the exact partition with dense tables; 4 = exact partition ending in compressed tables (occupancy
 * bits + rank + scores instead of dense per-group tables; what auto picks for AA k=6); 5 = ask for the quad kernel, 6 / 7 = force
 * the compressed / the dense table form behind the chunked pool); "debug_flags" (bits 0-4: timing experiments inside the scoring
 * kernels, results wrong; 5: no first chunks by position; 6: every host wait of a call kept -- no device-side counts, no
 * allocations from estimates, no key list inside the scoring call; 7 / 8: the quad kernel's workgroups never / always draw
 * their tiles; 9 / 10: the dense key-major writer with tile-by-tile / with line-cut stores whatever the group count; 11: 128-KB
 * table slices reduced by the workgroup-per-slice kernel instead of the persistent one; 12 / 13: the compressed key-major writer
 * per key block / per run of key blocks whatever the group count), "debug_pool_chunks", "debug_pool_limit_bytes",
 * "debug_wg_chunks2", "debug_rounds", "debug_kmc_pass" (groups per pass of the compressed key-major writer),
 * "debug_prefix_mats" (matrices per workgroup of the prefix-sum kernel: 1, 2, 4, 8; 0 = by the matrix count) (diagnostics and tests only).
 * Every variant yields identical results.  Returns IPKGPU_ERR_INVALID for unknown names.
 *
 * Calls on one context are synchronous, but from its second scoring call on a context waits on its stream ONCE per key-major
 * call: sizes it used to read back mid-call (chunks drawn, big-list windows, entry total) are taken from the previous call
 * as estimates and checked at that one wait; a wrong estimate costs a repeated pass, never a wrong result. */
int ipkgpu_set_option(ipkgpu_ctx* ctx, const char* name, int64_t value);

/* ---- host helpers (no GPU needed) ------------------------------------------------------ */

/* log10 of IPK's score threshold: db_builder.cpp:640 `std::log10(score_threshold(omega, k))`.
 * score_threshold is un-vendored i2l code; documented as (omega/sigma)^k
 * (docs/source/usage.rst:224-229) and evaluated here in float: log10f(powf(omega/sigma, k)). */
float ipkgpu_log_threshold(float omega, uint32_t sigma, uint32_t k);

/* Bits per symbol of the packed k-mer code: i2l::bit_length<seq_type>() as used at
 * pk_compute.cpp:99 (sigma 4 -> 2, sigma 20 -> 5).  0 for unsupported sigma. */
uint32_t ipkgpu_bits_per_symbol(uint32_t sigma);

/* ipk::kmer_batch -- branch_group.cpp:104-107. */
size_t ipkgpu_kmer_batch(uint32_t key, size_t n_ranges);

/* Largest supported k for an alphabet (DNA: 14, AA: 6); 0 for unsupported sigma. */
uint32_t ipkgpu_max_k(uint32_t sigma);

/* ---- the hot path ---------------------------------------------------------------------- */

/*
 * Replaces the body of the group loop, db_builder.cpp:606-625 + explore_group :629-698
 * (RAM mode), for a batch of ghost-node matrices.
 *
 *   logp       n_mats site-major matrices [mat][site][state] of float32 log10 posteriors --
 *              exactly the values raxmlng_reader::read_node produces (ar.cpp:257-260; AA columns
 *              already in IPK order, ar.cpp:232-234).  Host pointer.
 *   mat_group  [n_mats] branch id of each matrix = original post-order id
 *              (explore_kmers: _extended_mapping.at(node_group[0]), db_builder.cpp:613).
 *              Matrices with equal ids form one group (X0, X1 of a branch) in first-seen order.
 *   k, log_eps k-mer length and log10 score threshold (db_builder.cpp:640).
 *
 * Result (ipkgpu_result_* accessors): for every distinct branch id, the max-reduced set
 * {(key, score)} identical to group_map after explore_group (db_builder.cpp:685), keys packed
 * bits-per-symbol as at pk_compute.cpp:96-104, sorted ascending within the group (the
 * reference's order is hash order), plus the number of scored phylo-k-mers (the reference's
 * `count`, db_builder.cpp:664).
 *
 * Errors: IPKGPU_ERR_INVALID if sigma not in {4, 20}, k < 2, k > ipkgpu_max_k(sigma) or
 * sites < k (the reference reads out of range there, window.cpp:164-182; we reject).
 */
int ipkgpu_score_groups(ipkgpu_ctx* ctx, const float* logp, uint32_t n_mats, uint32_t sites,
                        uint32_t sigma, const uint32_t* mat_group, uint32_t k, float log_eps,
                        ipkgpu_result** out);

/* Same, with `logp` already resident in device memory (the measured configuration: no PCIe in
 * the timed region).  mat_group stays a host pointer.  The call returns after the device work
 * has completed; results stay in device memory until an accessor asks for a host copy. */
int ipkgpu_score_groups_device(ipkgpu_ctx* ctx, const float* logp_dev, uint32_t n_mats,
                               uint32_t sites, uint32_t sigma, const uint32_t* mat_group,
                               uint32_t k, float log_eps, ipkgpu_result** out);

/* KEEP_POSITIONS flavour (`ipk-aa-pos`: db_builder.cpp:655-662,687-689; branch_group.cpp:73-86): every kept score
 * carries the position (window start, window::get_position) of the window that produced it; on equal scores the
 * window processed first (group's matrices in input order, then ascending start) keeps its place, as `put`
 * replaces only on a strictly larger score.  Group-major output only; uses the global-atomic max-reduce with
 * 64-bit table entries.  ipkgpu_result_positions() gives the positions aligned with keys/scores. */
int ipkgpu_score_groups_positions(ipkgpu_ctx* ctx, const float* logp, uint32_t n_mats, uint32_t sites,
                                  uint32_t sigma, const uint32_t* mat_group, uint32_t k, float log_eps,
                                  ipkgpu_result** out);
const uint32_t* ipkgpu_result_positions(ipkgpu_result* r);   /* NULL unless produced by the call above */

/* ---- result accessors -------------------------------------------------------------------- */

uint32_t ipkgpu_result_num_groups(const ipkgpu_result* r);
/* [num_groups] branch ids in first-seen order (the order explore_kmers appends groups). */
const uint32_t* ipkgpu_result_group_ids(const ipkgpu_result* r);
/* [num_groups + 1] CSR offsets into keys/scores. */
const uint64_t* ipkgpu_result_offsets(const ipkgpu_result* r);
/* Total scored phylo-k-mers (sum over windows of |DCLA result|; db_builder.cpp:664,697). */
uint64_t ipkgpu_result_emitted(const ipkgpu_result* r);
/* Host copies (made on first use; NULL on failure) and the device-resident arrays. */
const uint32_t* ipkgpu_result_keys(ipkgpu_result* r);
const float* ipkgpu_result_scores(ipkgpu_result* r);
const uint32_t* ipkgpu_result_keys_device(const ipkgpu_result* r);
const float* ipkgpu_result_scores_device(const ipkgpu_result* r);
/* HIP-event timings of the call that produced r (milliseconds; see IPKGPU_T_*). */
double ipkgpu_result_time_ms(const ipkgpu_result* r, int which);
void ipkgpu_result_free(ipkgpu_result* r);

/* ---- key-major database parts and the k-mer-keyed merge (multi-GPU exchange step) ---------------- */

/*
 * Same scoring pass as ipkgpu_score_groups_device, but the result is delivered KEY-major and split
 * by owner -- the device analogue of the insertion loop at db_builder.cpp:685-694
 * (`_phylo_kmer_db.unsafe_insert(kmer, {branch, score})`, group after group) and of the on-disk
 * path's k-mer-keyed partition (kmer_batch, branch_group.cpp:104-107; merge_batch :45-70).
 *
 * Owner of a k-mer = dense_code % n_owners, where dense_code is the base-sigma value of the k-mer
 * (for DNA this IS the packed key, so owner = kmer_batch(key, n_owners)).  For owner o, key slot q
 * stands for dense code q * n_owners + o; slots = ceil(sigma^k / n_owners) per owner (zero padded).
 *
 *   counts   u32 [n_owners][slots]   number of (branch, score) entries of the key
 *   entries  {u32 branch, f32 score} owner-major, then ascending key, then group (first-seen) order
 *   owner_offsets  u64 [n_owners + 1] (host) entry offset of every owner's block
 *
 * With n_owners = world size, block o is what this rank sends to rank o (all-to-all over RCCL);
 * ipkgpu_merge_parts on the receiver concatenates, per key, the blocks of ranks 0..P-1.
 */
int ipkgpu_score_groups_keymajor_device(ipkgpu_ctx* ctx, const float* logp_dev, uint32_t n_mats, uint32_t sites,
                                        uint32_t sigma, const uint32_t* mat_group, uint32_t k, float log_eps,
                                        uint32_t n_owners, ipkgpu_parts** out);
uint32_t ipkgpu_parts_num_owners(const ipkgpu_parts* p);
uint64_t ipkgpu_parts_slots(const ipkgpu_parts* p);
const uint32_t* ipkgpu_parts_counts_device(const ipkgpu_parts* p);
const void* ipkgpu_parts_entries_device(const ipkgpu_parts* p);
const uint64_t* ipkgpu_parts_owner_offsets(const ipkgpu_parts* p);
uint64_t ipkgpu_parts_emitted(const ipkgpu_parts* p);
double ipkgpu_parts_time_ms(const ipkgpu_parts* p, int which);
void ipkgpu_parts_free(ipkgpu_parts* p);

/*
 * Merges, for one owner, the blocks received from n_sources ranks (or produced by n_sources batches):
 *   counts_dev      u32 [n_sources][slots] (device), source-major
 *   entries_dev     the sources' entry blocks (device); source s starts at entry source_offsets[s]
 *   source_offsets  u64 [n_sources] (host)
 * Result: this owner's shard of the phylo-k-mer database -- keys (IPK bit-packed codes, ascending),
 * key_offsets [num_keys + 1], entries {branch, score} with each key's entries in source order then
 * group order, i.e. the order the reference appends them (db_builder.cpp:606-618,685-694).
 */
int ipkgpu_merge_parts(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t owner, uint32_t n_owners,
                       uint32_t n_sources, const uint32_t* counts_dev, const void* entries_dev,
                       const uint64_t* source_offsets, ipkgpu_db** out);
/* Like ipkgpu_merge_parts, with one pointer pair per source instead of one base and offsets: counts_dev[s] = the source's
 * counts row [slots] (device), entries_dev[s] = its entry block (device).  Both pointer arrays live in HOST memory. */
int ipkgpu_merge_parts_ptrs(ipkgpu_ctx* ctx, uint32_t sigma, uint32_t k, uint32_t owner, uint32_t n_owners, uint32_t n_sources,
                            const uint32_t* const* counts_dev, const void* const* entries_dev, ipkgpu_db** out);

/* ---- the exchange step itself: RCCL over xGMI, inside the library ------------------------------------------------
 * One process per GPU.  Rank 0 draws an id (ipkgpu_comm_unique_id) and hands its 128 bytes to the other ranks by any means
 * (MPI, a file, torch.distributed); every rank then calls ipkgpu_comm_init.  Per range ("piece") of its branch groups a rank
 * calls ipkgpu_score_groups_keymajor_device(..., n_owners = world) and ipkgpu_exchange_begin, which enqueues block o -> rank o
 * (grouped ncclSend/ncclRecv on the communicator's own stream) and returns, so the transfer runs under the next piece's
 * scoring; ipkgpu_exchange_merge waits for the pieces and merges (rank, piece)-ordered sources into this rank's shard.
 * All ranks must use the same number of pieces.  RCCL is loaded at run time: IPKGPU_ERR_NODEVICE if it is not there.
 * Failure must be symmetric, or the healthy ranks wait for ever inside a collective: everything that can fail on one rank
 * alone (loading RCCL, the exchange stream and buffers) is done by ipkgpu_comm_prepare, BEFORE the collective
 * ncclCommInitRank inside ipkgpu_comm_init -- the ranks agree on the prepare results first (an all-reduce by whatever carried
 * the id) and enter ipkgpu_comm_init only if all succeeded.  A rank whose ipkgpu_exchange_begin fails aborts the communicator
 * (ncclCommAbort), which makes its peers' pending transfers fail instead of hang; it should then exit non-zero.
 * CPU analogue: branch_group.cpp:45-70,104-107 (merge_batch, kmer_batch), db_builder.cpp:392-458 (merge_stage2). */
typedef struct ipkgpu_xfer ipkgpu_xfer;
int ipkgpu_comm_available(void);                          /* IPKGPU_OK if RCCL can be loaded in this process (no GPU call) */
int ipkgpu_comm_unique_id(uint8_t* id128);
int ipkgpu_comm_prepare(ipkgpu_ctx* ctx, int world);       /* the local, non-collective half of the set-up (idempotent) */
int ipkgpu_comm_init(ipkgpu_ctx* ctx, const uint8_t* id128, int rank, int world);
int ipkgpu_comm_rank(const ipkgpu_ctx* ctx);
int ipkgpu_comm_world(const ipkgpu_ctx* ctx);
int ipkgpu_exchange_begin(ipkgpu_ctx* ctx, ipkgpu_parts* parts, ipkgpu_xfer** out);     /* parts stay alive until the merge */
int ipkgpu_exchange_merge(ipkgpu_ctx* ctx, ipkgpu_xfer* const* xfers, uint32_t n_pieces, uint32_t sigma, uint32_t k, ipkgpu_db** out,
                          double* exposed_ms /* optional: time spent waiting for transfers */);
double ipkgpu_xfer_exposed_ms(const ipkgpu_xfer* x);
void ipkgpu_xfer_free(ipkgpu_xfer* x);

/* Single-GPU shortcut (n_owners == 1): the parts already ARE the database; produces the key list and
 * MOVES the entry array out of `parts` (which stays valid for its counts/timings, entries become NULL). */
int ipkgpu_db_from_parts(ipkgpu_ctx* ctx, ipkgpu_parts* parts, uint32_t sigma, uint32_t k, ipkgpu_db** out);
uint64_t ipkgpu_db_num_keys(const ipkgpu_db* d);
uint64_t ipkgpu_db_num_entries(const ipkgpu_db* d);
/* host copies (made on first use): keys u32[num_keys], key_offsets u64[num_keys+1], entries u32[num_entries][2] */
const uint32_t* ipkgpu_db_keys(ipkgpu_db* d);
const uint64_t* ipkgpu_db_key_offsets(ipkgpu_db* d);
const uint32_t* ipkgpu_db_entries(ipkgpu_db* d);
const uint32_t* ipkgpu_db_keys_device(const ipkgpu_db* d);
const uint64_t* ipkgpu_db_key_offsets_device(const ipkgpu_db* d);
const void* ipkgpu_db_entries_device(const ipkgpu_db* d);
double ipkgpu_db_time_ms(const ipkgpu_db* d);
void ipkgpu_db_free(ipkgpu_db* d);

/* ---- "next" row n1: MIF0 filter values and k-mer order (filter.cpp:55-119, db_builder.cpp:254-284) -- */

/* IPK's score threshold itself (un-vendored i2l::score_threshold; assumed powf(omega/sigma, k)). */
float ipkgpu_score_threshold(float omega, uint32_t sigma, uint32_t k);

/*
 * mif0_filter::calc_filter_values over a database shard, then the order of db_builder.cpp:284
 * (ascending filter value; ties -- unspecified in the reference -- by ascending key).
 *   total_num_groups  N = node count of the original tree (db_builder.cpp:261)
 *   threshold         score_threshold(omega, k), NOT its log (db_builder.cpp:260)
 * Filter values are per k-mer, so a shard is filtered independently of the other owners.
 * Double arithmetic on the device, the entries' terms ADDED IN ENTRY ORDER (the association of the reference's two
 * sequential loops, filter.cpp:60-119): the float filter values and the k-mer order equal a sequential host evaluation;
 * only the last bit of the device's pow / log2 could differ from the host's libm.
 */
int ipkgpu_db_filter_mif0(ipkgpu_ctx* ctx, ipkgpu_db* db, uint64_t total_num_groups, float threshold);
/* host copies: filter value per k-mer (as the float i2l::kmer_fv stores, and in double), and the
 * positions of the k-mers in filter order: keys[order[0]] is written first */
const float* ipkgpu_db_filter_values(ipkgpu_db* d);
const double* ipkgpu_db_filter_values_f64(ipkgpu_db* d);
const uint32_t* ipkgpu_db_filter_order(ipkgpu_db* d);
const float* ipkgpu_db_filter_values_device(const ipkgpu_db* d);
const uint32_t* ipkgpu_db_filter_order_device(const ipkgpu_db* d);
double ipkgpu_db_filter_time_ms(const ipkgpu_db* d);

/* ---- "next" row n3: RAxML-ng ancestral-probabilities loader (host, multi-threaded) -------------------- */

typedef struct ipkgpu_ar ipkgpu_ar;

/* raxmlng_reader (ar.cpp:144-188): memory-maps `<prefix>.raxml.ancestralProbs` (TSV, one header line, rows
 * `Node \t Site \t State \t p_1 .. p_sigma`, contiguous per node) and indexes the node blocks. */
int ipkgpu_ar_open(const char* path, uint32_t sigma, ipkgpu_ar** out);
void ipkgpu_ar_close(ipkgpu_ar* ar);
const char* ipkgpu_ar_last_error(void);           /* message of the last failing ipkgpu_ar_* call of this thread */
uint32_t ipkgpu_ar_num_nodes(const ipkgpu_ar* ar);
uint32_t ipkgpu_ar_sites(const ipkgpu_ar* ar);    /* rows of the first node's block */
const char* ipkgpu_ar_node_label(const ipkgpu_ar* ar, uint32_t i);   /* labels in order of first appearance */
int64_t ipkgpu_ar_find(const ipkgpu_ar* ar, const char* label);      /* -1 if absent */
/* raxmlng_reader::read_node (ar.cpp:200-270) for n nodes at once: out[i] = [sites][sigma] float32 log10
 * posteriors of node node_idx[i] (AA columns permuted to IPK order), ready for ipkgpu_score_groups.
 * n_threads = 0 uses all host cores. */
int ipkgpu_ar_read_nodes(ipkgpu_ar* ar, const uint32_t* node_idx, uint32_t n, float* out, uint32_t n_threads);

/* ---- "next" row n3 (second half): reference tree, ghost nodes, node mapping (host) -------------------------- */

typedef struct ipkgpu_tree ipkgpu_tree;
typedef struct ipkgpu_ghost_plan ipkgpu_ghost_plan;

/* Own newick reader (i2l::io::load_newick is un-vendored): children in file order, post-order ids from 0, quoted labels
 * and [comments] accepted.  ipkgpu_tree_last_error(): message of the last failing ipkgpu_tree_* / ipkgpu_ghost_plan_*
 * call of this thread. */
int ipkgpu_tree_parse(const char* newick, ipkgpu_tree** out);
int ipkgpu_tree_load(const char* path, ipkgpu_tree** out);
void ipkgpu_tree_free(ipkgpu_tree* t);
const char* ipkgpu_tree_last_error(void);
uint32_t ipkgpu_tree_num_nodes(const ipkgpu_tree* t);
uint32_t ipkgpu_tree_num_leaves(const ipkgpu_tree* t);
int ipkgpu_tree_is_rooted(const ipkgpu_tree* t);                      /* root has exactly two children */
const char* ipkgpu_tree_label(const ipkgpu_tree* t, uint32_t postorder_id);
int64_t ipkgpu_tree_parent(const ipkgpu_tree* t, uint32_t postorder_id);   /* post-order id of the parent, -1 for the root */
double ipkgpu_tree_branch_length(const ipkgpu_tree* t, uint32_t postorder_id);
const char* ipkgpu_tree_newick(ipkgpu_tree* t);                       /* owned by the tree, valid until the next call */
/* The database header's tree index (db_builder.cpp:192-197): per node in post-order, the size of its subtree and the
 * total branch length below it.  Arrays of ipkgpu_tree_num_nodes() elements. */
int ipkgpu_tree_index(const ipkgpu_tree* t, uint32_t* num_nodes, double* subtree_length);
/* tree_extender::extend (extended_tree.cpp:76-150): ghost nodes <counter>_X0 / _X1 (+ dummy leaves _X2 / _X3) on every
 * non-root branch, counter from node_count + 1; the result remembers ghost label -> original post-order id. */
int ipkgpu_tree_extend(const ipkgpu_tree* original, ipkgpu_tree** extended);
/* reroot_tree (extended_tree.cpp:186-205), applied to the AR tree when the reference tree is rooted (main.cpp:172-178). */
int ipkgpu_tree_reroot(ipkgpu_tree* t);
/* get_ghost_ids + group_ghost_ids (db_builder.cpp:495-553) + map_nodes (ar.cpp:790-834): the ghost nodes in the order
 * explore_kmers scores them, each with its label in the AR output and its branch id = mat_group of ipkgpu_score_groups.
 * strategy: 0 both, 1 inner only (_X0), 2 outer only (_X1).  ar_tree may be NULL (AR labels = extended labels). */
int ipkgpu_ghost_plan_make(const ipkgpu_tree* original, const ipkgpu_tree* extended, const ipkgpu_tree* ar_tree, int strategy,
                           ipkgpu_ghost_plan** out);
void ipkgpu_ghost_plan_free(ipkgpu_ghost_plan* p);
uint32_t ipkgpu_ghost_plan_size(const ipkgpu_ghost_plan* p);
const char* ipkgpu_ghost_plan_ext_label(const ipkgpu_ghost_plan* p, uint32_t i);
const char* ipkgpu_ghost_plan_ar_label(const ipkgpu_ghost_plan* p, uint32_t i);
const uint32_t* ipkgpu_ghost_plan_branches(const ipkgpu_ghost_plan* p);

/* ---- "next" row n2: the database file (db_builder.cpp:145-146,176-177,297-306,323-327) ------------------------ */

/* Header fields in the order of i2l::ipk_header as db_builder.cpp:297-305 fills it. */
typedef struct ipkgpu_db_header {
    const char* sequence_type;            /* seq_type::name: "DNA" | "AA" */
    uint64_t tree_index_size;             /* nodes of the original tree */
    const uint32_t* tree_num_nodes;       /* [tree_index_size] (ipkgpu_tree_index) */
    const double* tree_subtree_length;    /* [tree_index_size] */
    const char* newick;                   /* the original tree */
    uint64_t kmer_size;
    float omega;
} ipkgpu_db_header;

/* save_header + save_phylo_kmer for every k-mer in filter order (db_builder.cpp:297-306,323-327), streamed from device
 * memory: the records are packed on the GPU in pieces, copied through pinned buffers and written while the next piece is
 * being packed.  Needs ipkgpu_db_filter_mif0 first.  Byte layout: ipk_amd/csrc/ipk_format.hpp -- i2l and Boost are
 * un-vendored, so the layout is a reconstruction (Boost binary_oarchive primitives) and NOT pinned against a real .ipk. */
int ipkgpu_db_write(ipkgpu_ctx* ctx, ipkgpu_db* db, const ipkgpu_db_header* header, const char* path, uint64_t* bytes_written);
/* The same serialiser over host arrays (merged shards of several GPUs, or a filter computed on the host):
 * entries u32 [n][2] = (branch, score bits); order = positions of the k-mers in output order (NULL: as stored). */
int ipkgpu_db_write_host(const ipkgpu_db_header* header, uint64_t n_keys, const uint32_t* keys, const uint64_t* key_offsets,
                         const uint32_t* entries, const float* filter_values, const uint32_t* order, const char* path,
                         uint64_t* bytes_written);
/* The positioned database of ipk-aa-pos (KEEP_POSITIONS: db_builder.cpp:655-662,687-689; branch_group.cpp:73-86): as
 * ipkgpu_db_write_host, with positions[n] = the window position that goes with every entry's score (ipkgpu_score_groups_positions),
 * the header's positions flag set and entries of (branch, score, position).  The position's width in the file (u16) is a guess like
 * the rest of the layout; positions beyond 65535 are refused. */
int ipkgpu_db_write_host_positions(const ipkgpu_db_header* header, uint64_t n_keys, const uint32_t* keys, const uint64_t* key_offsets,
                                   const uint32_t* entries, const uint32_t* positions, const float* filter_values, const uint32_t* order,
                                   const char* path, uint64_t* bytes_written);
const char* ipkgpu_db_write_last_error(void);
/* The database file of a multi-GPU build -- the role of merge_stage2 (db_builder.cpp:392-458: batch files opened together,
 * a priority queue on the filter value hands out the k-mer to append next).  Every rank writes ITS shard (the k-mers it owns,
 * filter values computed, in its filter order) as a database file of its own with ipkgpu_db_write / ipkgpu_db_write_host
 * (any header: only the totals are read back); one rank then merges the P shard files by (filter value, key) into `path`
 * under header `h`, streaming through bounded buffers: resident memory does not depend on the number of entries, and the
 * result equals, byte for byte, the file one GPU writes for the same input.  Host code, no GPU needed. */
int ipkgpu_db_merge_files(const ipkgpu_db_header* h, const char* const* shard_paths, uint32_t n_shards, const char* path,
                          uint64_t* total_kmers, uint64_t* total_entries, uint64_t* bytes_written);
const char* ipkgpu_db_merge_last_error(void);
/* seconds of the last ipkgpu_db_write of this context: 0 = total, 1 = device packing + copies (waited for), 2 = file writes */
double ipkgpu_db_write_time_s(const ipkgpu_ctx* ctx, int which);
/* The protocol version the writers put behind the archive preamble, followed (after the sequence type) by the positions flag --
 * what phylo_kmer_db::version() / positions_loaded() answer for a loaded database (tools/src/diff.cpp:41-46,137-145).  Position,
 * width and value are guesses (ipk_format.hpp); IPKGPU_IPK_PROTOCOL_VERSION in the environment sets the value, 0 = neither field
 * is written.  Readers of this library's files (ipkgpu_db_merge_files, the tests' parser) ask here what to expect. */
uint32_t ipkgpu_db_protocol_version(void);

#ifdef __cplusplus
}
#endif
#endif /* IPKGPU_H */
