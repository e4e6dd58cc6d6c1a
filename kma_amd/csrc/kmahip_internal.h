// kmahip_internal.h -- shared between the host-side loader and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
#include "../../include/kmahip.h"

#define KMAHIP_EMPTY_VI 0xFFFFFFFFu
#define KMAHIP_BUCKET_SLOTS 4
#define KMAHIP_N_COUNTERS 24
#define KMAHIP_KBITS_MUL 0x85EBCA6Bu      // device counter words per workspace

// Probe table in HBM: open hashing over 32-byte buckets of 4 (key, position)
// slots. bucket(key) = (key * GOLD) >> (32 - nb_log2); a key lives in the first
// bucket at or after its home bucket (cyclic) that had a free slot at build
// time, so a lookup stops at the first bucket holding either the key (hit) or
// an empty slot (miss). This replaces the reference's three dependent gathers
// exist[] -> key_index[] -> value_index[] (hashmapkma.c:149-178) by one 32-byte
// gather in the common case.
struct DevDB {
	uint32_t DB_size;
	uint32_t kmersize;
	uint32_t mlen;
	uint32_t nb_log2;
	uint32_t values_u16;          // 1: values16 valid, 0: values32
	const uint2 *slots;           // (1 << nb_log2) * 4 slots of (key, gpos): gpos = one position of the k-mer in `cat`
	const uint64_t *cat;          // all templates back to back, 2 bit per base (+ pad words)
	const uint32_t *vs_id;        // per position of `cat`: value-list offset of the k-mer starting there, or
	                              // KMAHIP_EMPTY_VI where no k-mer of one template starts (its last k-1 bases)
	const int64_t *cat_off;       // DB_size + 1: first position of template t in `cat` (cat_off[DB_size] = total bases)
	const uint16_t *values16;     // [cnt, t1..tcnt] lists, offsets = value_index of the index file
	const uint32_t *values32;
	const int32_t *tlen;          // DB_size, tlen[0] = kmerindex
	const uint64_t *tseq;         // 2-bit template store (.seq.b image + pad)
	const int64_t *tseq_off;      // DB_size + 1 word offsets
	// per-template position index (replaces the lazily built HashMapCCI,
	// hashmapcci.c:470-505): linear-probing table of (kmer, val) per template;
	// val > 0: the single 1-based position of the k-mer; val < 0: -(o + 1) where
	// tpos_dups[o] = count followed by the ascending 1-based positions; val == 0
	// empty. The poly-A k-mer 0 is never indexed (hashmapcci.c:414-417).
	// presence bits of the k-mers of the probe table (small databases only, else NULL): bit ((key * KBITS_MUL) >> kbits_shift).
	// 2 MiB at most, so it stays in the 4 MiB L2 of every XCD, where a gather costs a quarter of one into the 67 MB table;
	// consulted where a miss is the likely answer (wrong-strand prefilter probes)
	const uint32_t *kbits;
	uint32_t kbits_shift;
	const uint2 *tpos_slots;
	const int64_t *tpos_off;      // DB_size + 1 slot offsets
	const uint32_t *tpos_shift;   // DB_size: 32 - log2(table size)
	const int32_t *tpos_dups;
	// everything the seeding kernel needs to know about a template in one 32-byte record (two 16-byte gathers instead of four
	// scattered ones): x = {tseq_off lo, hi, tpos_off lo, hi}, y = {tlen, tpos_shift, 0, 0}
	const uint4 *tmeta;           // 2 * DB_size
};

struct kmahip_db {
	DevDB dev;                    // device pointers
	kmahip_db_info info;
	int device;
	std::vector<void *> allocs;   // device allocations owned by the db
	// host copies needed by host-side stages
	std::vector<int32_t> h_tlen;
	std::vector<int64_t> h_cat_off;
	std::string prefix;           // index prefix (the `.name` file is read on demand by the text writers)
	std::vector<std::string> h_names;
	std::vector<uint64_t> h_tseq;     // <prefix>.seq.b, read when the `.aln` writer first asks (pipeline.hip: load_tseq)
	std::vector<int64_t> h_tseq_off;
};

// per-call scratch, grown on demand
struct kmahip_ws {
	kmahip_db *db;
	int64_t cap_reads;
	// per strand item (2 per read)
	int32_t *item_score;
	int32_t *item_n;
	int64_t *item_off;
	// candidate pool
	int32_t *pool;
	int64_t pool_cap;
	int64_t pool_scale;       // pool = cap_reads * 16 * pool_scale ints; doubled after an overflow
	int mem_scale;            // MEM slots per (read, template) = the usual 64 (reads up to 1 kb) x this; raised by the runs when a read
	                          // full of repeats carries more (status 3), 0 = 1
	// counters (16): [0] pool top, [1] status, [2] n_overflow, [3] probes, [4] value elems, [5] active strands,
	// [6] hash probes, [7] pair pool top, [8] active strand items, [9] prefilter probes
	unsigned long long *counters;
	int64_t *overflow_items;
	int64_t *active_items;    // strand items that passed the prefilter (device-wide compaction)
	int stats_on;
	int timing_on;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> *events;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> *events2;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> *events3;   // prefilter kernel
	std::vector<std::pair<hipEvent_t, hipEvent_t>> *events4;   // seeding kernel
	// paired input in the default mode (kmahip_ws_set_pe_chain): singly loaded reads of a paired stream go to the chain finder
	int pe_chain_on;
	kmahip_chain_params pe_chain;
	// paired-end stage 2
	int32_t *pool_sc, *ppool;
	void *pe_rec;
	int64_t pe_cap;
	// align stage
	int32_t *a_s32;
	uint64_t *a_s64;
	int64_t a_lanes;
	int a_mem_cap, a_ncols;
	void *a_task;
	int64_t a_task_cap;
	int *a_xq;                    // spill queues of deferred DP problems (long reads), see align.hip
	size_t a_xq_bytes;
	hipStream_t a_side;           // the general kernel over the handed-on tasks runs here, beside the class queues (align.hip)
	hipEvent_t a_ev[2];           // fork / join of that stream
	void *a_priv;             // private copies of the ConClave vectors (reduce_reads_kernel)
	int64_t a_priv_cap;
	// trace stage (3c) scratch
	int32_t *t_s32;
	uint8_t *t_E;
	int64_t t_lanes;
	int32_t *t_queue;            // reads the first trace pass put off
	int64_t t_queue_cap;
	int t_max_len, t_mem_cap;
	// pile-up stage (3c) scratch and results
	uint32_t *p_counts;
	int32_t *p_chain, *p_seg, *p_vals;
	void *p_nodes;
	uint64_t *p_keys;
	int64_t *p_rank;
	int64_t p_total, p_node_cap, p_reads_cap, p_kept, p_nodes_used, p_ent_cap;
	// long-read trace pipeline (longtrace.hip): per-wavefront MEM arrays, per-pass pools, queues, scratch, counters
	void *lt_buf[20];
	size_t lt_bytes[20];
	unsigned long long lt_stats[4];   // of the last call: DP problems, DP cells, MEMs chained, reads
	// slow-path dense scratch
	int32_t *dense;
	int64_t dense_slots;
	// scan workspace for CSR offsets
	int64_t *blk_sums;
	int64_t blk_cap;
	// staging for host calls
	void *stage[8];
	size_t stage_bytes[8];
};

// an anchor of the default mode's chain finder (KmerAnker, kmeranker.h:25-34): a maximal run of k-mer starts with one value list.
// Built by chain_anchor_kernel (scan.hip) or by the lane-per-read kernel of chain.hip, chained and taken apart in chain.hip.
struct KmaAnk { int score, weight, score_len, len_len; unsigned start, end; uint32_t values; int descend; };

void kmahip_set_error(const char *fmt, ...);
int kmahip_cmp(bool t, bool q);          // conclave.hip: the reference's `cmp` (or / and / true, kmahip_set_cmp)
#define HIP_TRY(expr) do { hipError_t e__ = (expr); if(e__ != hipSuccess) { \
	kmahip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); return KMAHIP_EDEVICE; } } while(0)

int kmahip_launch_scan_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                          const kmahip_params *p, kmahip_cands *out, hipStream_t stream);
int kmahip_launch_align_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                           const kmahip_params *p, kmahip_hits *out, hipStream_t stream);
int kmahip_launch_trace(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                        const uint8_t *tmpl_ok, const kmahip_params *p, kmahip_traces *out, hipStream_t stream);
int kmahip_trace_reserve(kmahip_ws *ws, int max_len, int64_t n);          // align.hip: the traceback scratch, ahead of time
int kmahip_launch_longtrace(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *tmpl, int tmpl_all, const int32_t *rc_in,
                            const uint8_t *tmpl_ok, int one2one, const kmahip_params *p, kmahip_traces *out, int32_t *rc_out, hipStream_t stream, int score_mode = 0);
// anchors of every strand that passes the prefilter, for reads without N's and up to 288 k-mer starts (the others get slow[read] = 1):
// a_n[2 r + strand] anchors at pool + a_off[2 r + strand]; cnt[0] = anchors written (may exceed pool_cap: repeat with a larger pool)
// the fragment rows of a run whose items (reads, records, fragments) and headers are in HBM: ordered, measured and formatted on the
// device, compressed and written by the host's threads (session.hip)
int kmahip_frag_write_dev(kmahip_db *db, const kmahip_reads *W, const char *d_names, const int64_t *d_name_off, const int64_t *d_name_idx, const int32_t *d_rc,
                          const int32_t *d_tmpl, const int32_t *d_nhits, const int32_t *d_stats, const int64_t *d_rank, int64_t max_frag, const char *path,
                          int64_t text_chunk, char **pinned, int64_t *n_rows_out, int order = 0, struct KmaFragSink *sink = nullptr);
// a fragment file that stays open over several kmahip_frag_write_dev calls (each appends its rows; kmahip_frag_sink_close ends the file):
// the compressing threads go on with a call's text after the call has returned, in the caller's three pinned buffers (`pinned` is then
// required, and the same for every call)
struct KmaFragSink;
KmaFragSink *kmahip_frag_sink_open(const char *path);
int kmahip_frag_sink_close(KmaFragSink *sink);

// stage 2 of the default mode on a batch that is in HBM, its records as a batch of their own in stream order (pipeline.hip). The
// arrays live in a block of their own until kmahip_chain_records_free.
struct KmaChainRecs {
	int64_t m = 0, n_T = 0;
	kmahip_reads d{};              // the records: the read, or its reverse complement where the record prints that; q_start / q_end set
	kmahip_cands c{};              // their template lists
	int64_t *o_read = nullptr;     // the read of the batch a record comes from
	int32_t *o_emit = nullptr;     // 1 = the record holds the reverse complement
	void *block = nullptr;
};
int kmahip_chain_records_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *dR, const kmahip_params *p, const kmahip_chain_params *cp, KmaChainRecs *out);
void kmahip_chain_records_free(KmaChainRecs *r);

int kmahip_launch_chain_anchors(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, KmaAnk *pool, int64_t pool_cap,
                                int64_t *a_off, int32_t *a_n, uint8_t *slow, unsigned long long *cnt, hipStream_t stream);
int kmahip_db_load_names(kmahip_db *db);       // fragout.hip: fills db->h_names from <prefix>.name
double kmahip_p_chisqr(long double q);      // stdstat.c:136-147 (conclave.hip)
int kmahip_launch_scan_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                          kmahip_pe_recs *out, hipStream_t stream);
int kmahip_launch_align_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                           const kmahip_params *p, kmahip_hits *out, int32_t *pe_kind, hipStream_t stream);

// the paired run on a batch that is in HBM already (session.hip -> pipeline.hip): batch->reads holds DEVICE arrays (sizes and
// max_len as usual), batch->pair the host's mate flags; the headers and the fragment writer's pinned text buffers are these
struct KmaPeDev {
	const char *d_names;
	const int64_t *d_name_off;
	char **h_text;
	int64_t text_chunk;
	int64_t *frag_rows;       // out: rows written to the fragment file (may be NULL)
};
int kmahip_run_pe_resident(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const KmaPeDev *pd, const kmahip_params *p, double evalue, int bcd,
                           int64_t max_frag, const char *frag_path, kmahip_run *out);
