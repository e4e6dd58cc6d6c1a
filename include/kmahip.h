/* kmahip.h -- C ABI of libkmahip.so: the MI355X (gfx950) implementation of KMA's
 * seed-and-extend mapping core (stage 2 "k-mer scan" and stage 3a "alignment
 * score"), batched.  Plain C types only; every function returns 0 on success or
 * a negative KMAHIP_E* code (never exit()s; the reference's ERROR() ->
 * exit(errno) convention, pherror.h:27, is left to the calling host program).
 *
 * What each entry point replaces in the reference (file:line into KMA 1.5.1):
 *   kmahip_db_open            hashMapKMA_load           hashmapkma.c:275-455
 *                             load_DBs_KMA + seq_indexes runkma.c:160-220
 *                             alignLoad_fly/hashMapCCI_load (lazy per template)
 *                                                        hashmapcci.c:470-505,616
 *   kmahip_scan_se[_dev]      save_kmers_batch body:    kmers.h:22, kmers.c:51-290
 *                             worker loop save_kmers_threaded savekmers.c:94-271
 *                             with kmerScan = save_kmers (savekmers.h:50,
 *                             savekmers.c:2442-3065) and hashMap_get
 *                             (hashmapkma.h:59, hashmapkma.c:149-178); the
 *                             result arrays carry the S2 record fields that
 *                             print_ankers writes (ankers.c:30-50)
 *   kmahip_align_se[_dev]     alnFrags_threaded body (alnfrags.c:2150-2294) with
 *                             alnFragsSE (alnfrags.c:1052-1218): KMA_score
 *                             (align.h:33, align.c:509-748), chainSeeds
 *                             (chain.h:39, chain.c:79-260), NW_score /
 *                             NW_band_score (nw.h:62-63, nw.c:642-1188),
 *                             update_Scores (updatescores.c:203-298)
 *   kmahip_map_se             both of the above on one staged batch (host buffers)
 *   kmahip_allreduce_scores   (new) SUM of alignment_scores/uniq_alignment_scores
 *                             across read shards before runConClave
 *                             (runkma.c:563-594); see INTEGRATION.md
 *
 * Ownership: the caller owns every host and device buffer it passes in; the
 * library owns the database image in HBM and its private workspaces.
 * Threading: one kmahip_db may be used from several host threads as long as
 * each call uses its own kmahip_ws (workspace) and stream.
 */
#ifndef KMAHIP_H
#define KMAHIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KMAHIP_OK            0
#define KMAHIP_EINVAL       -1  /* bad argument */
#define KMAHIP_EIO          -2  /* cannot read an index file */
#define KMAHIP_EFORMAT      -3  /* index variant not supported (megamap, k>16 keys, flag!=0) */
#define KMAHIP_ENOMEM       -4  /* host or device allocation failed */
#define KMAHIP_EDEVICE      -5  /* HIP runtime error (see kmahip_last_error) */
#define KMAHIP_EOVERFLOW    -6  /* an output capacity given by the caller was too small */

typedef struct kmahip_db kmahip_db;
typedef struct kmahip_ws kmahip_ws;

/* scoring constants: Penalties, penalties.h:22-33 (d is the 5x5 substitution
 * matrix built at kma.c:1307-1328) */
typedef struct kmahip_rewards {
	int32_t M, MM, U, W1, Wl, Mn, PE;
	int32_t d[5][5];
} kmahip_rewards;

/* run parameters (the subset of kma.c flags the path reads) */
typedef struct kmahip_params {
	kmahip_rewards rw;
	int32_t exhaustive;   /* -ex_mode */
	int32_t minlen;       /* -ml, default 16 */
	int32_t mq;           /* -mq, default 0 */
	double scoreT;        /* -mrs, default 0.5 */
	double mrc;           /* -mrc, default 0.0 */
	double minFrac;       /* 1.0 */
	int32_t ts;           /* -ts, default 0: the traceback aligner trims this many bases off the front of every seed of the best chain but
	                       * the first one when that starts the read (trimSeeds chain.c:493-528, called by KMA() align.c:413; KMA_score
	                       * has no such step); at least one base of a seed remains */
	int32_t apm;          /* paired reads: 0 = the pairing penalty of -apm p (save_kmers_penaltyPair savekmers.c:3572, alnFragsPenaltyPE
	                       * alnfrags.c:1596), 1 = union, -apm u and what `-ipe` means without -apm (kma.c:206: save_kmers_unionPair
	                       * savekmers.c:3367 with getF_Best / getR_Best, alnFragsUnionPE alnfrags.c:1220). The reference can set the two
	                       * stages apart (-pm x: save_kmers_pair only, -fpm x: alnFragsPE only, kma.c:437-465): bits 0-1 = stage 2 as
	                       * above, bits 4-5 = stage 3a + 1 (0: the same as stage 2, 1: p, 2: u). Stage 2 also takes 2 = forced pairing
	                       * (-apm f / -pm f: save_kmers_forcePair savekmers.c:3779 with getFirstForce / getSecondBestForce -- a couple on the
	                       * templates both mates hit on opposite strands at the best summed score, or no record at all) -- for kmahip_scan_pe
	                       * only: stage 3a of forced pairing (alnFragsForcePE) is not built */
} kmahip_params;

typedef struct kmahip_db_info {
	uint32_t DB_size;     /* templates + 1 (ids are 1-based) */
	uint32_t kmersize;
	uint64_t n_kmers;     /* distinct k-mers */
	uint64_t n_values;    /* elements in the value-list array */
	uint64_t hash_bytes;  /* bytes of the probe table in HBM */
	uint64_t total_bytes; /* all DB bytes resident in HBM */
	uint64_t tseq_words;  /* 2-bit template store, u64 words */
} kmahip_db_info;

/* A batch of 2-bit packed reads (CompDNA, compdna.h:23-30), CSR layout.
 * seq: u64 words, 32 bases per word MSB-first, N packed as A; each read is
 * followed by at least one readable pad word (getKmer_macro reads word+1,
 * stdnuc.h:27-30).  N: sorted N positions.  All pointers are HOST pointers for
 * the plain calls and DEVICE pointers for the *_dev calls. */
typedef struct kmahip_reads {
	int64_t n_reads;
	const uint64_t *seq;
	const int64_t *seq_off;   /* n_reads+1 word offsets */
	const int32_t *len;       /* n_reads */
	const int32_t *N;
	const int64_t *N_off;     /* n_reads+1 */
	int64_t seq_words;        /* total words in seq (host calls: bytes to stage) */
	int64_t N_total;
	int32_t max_len;          /* longest read in the batch (sizes the align scratch) */
	/* optional (NULL: whole reads): query bounds per read, [q_start[i], q_end[i]) in the coordinates of the read as stored --
	 * what a default-mode S2 record carries behind its header (qseqs.c:41-56). Stage 3a and the traceback look for seeds only
	 * from q_start on and let the last N-free stretch end at q_end (KMA_score align.c:534-540, KMA :249-254, anker_rc_comp
	 * :1029-1044; for a view of the reverse strand the bounds count from the other end, alnfrags.c:1113-1127). Same memory
	 * space as the other arrays of the call. */
	const int32_t *q_start;
	const int32_t *q_end;
} kmahip_reads;

/* Stage-2 result, one entry per read = the S2 record fields (ankers.c:30-50):
 * rc_flag = +-best k-mer score (negative: both strands tie), flag = 0 | 16
 * (16: the reverse-complemented read is the one passed on), T = candidate
 * template ids (negative id: reverse strand entry of a tie). Reads the
 * reference would not emit have T_off[i+1] == T_off[i]. */
typedef struct kmahip_cands {
	int32_t *rc_flag;     /* n_reads */
	int32_t *flag;        /* n_reads */
	int64_t *T_off;       /* n_reads + 1 */
	int32_t *T;           /* T_cap */
	int64_t T_cap;
} kmahip_cands;

/* Stage-3a result (what update_Scores keeps, updatescores.c:203-298 = one
 * frag_raw record per read).  Hits of read i are stored at
 * [T_off[i], T_off[i] + n_hits[i]) of tmpl/score/start/end, in candidate order,
 * so these arrays need the same capacity as kmahip_cands.T.  Reads whose two
 * strands tied in stage 2 (rc_flag < 0) get their strand per template from MEM
 * coverage (anker_rc_comp, align.c:993-1176); a negative tmpl = reverse strand.
 * alignment_scores / uniq_alignment_scores are the two u64[DB_size] ConClave
 * vectors (updatescores.c:228,276); the call ADDS into them (caller zeroes). */
typedef struct kmahip_hits {
	int32_t *n_hits;      /* n_reads */
	int32_t *best_score;  /* n_reads: best_read_score */
	int32_t *flag;        /* n_reads: stage-2 flag, |= 4 when unmapped after alignment */
	int32_t *tmpl;        /* signed template id */
	int32_t *score;
	int32_t *start;
	int32_t *end;
	uint64_t *alignment_scores;       /* DB_size, may be NULL */
	uint64_t *uniq_alignment_scores;  /* DB_size, may be NULL */
	int32_t *rc;          /* n_reads (records), may be NULL. bit 0: the fragment update_Scores* files for this record is the
	                       * reverse complement of the ORIGINAL read (single end: stage 2's flag & 16; paired: a pair with a
	                       * reverse-strand candidate is turned and only turned back, flag toggled, when its first kept template
	                       * is positive, alnfrags.c:1629-1643, 1807-1822 -- the flag alone does not tell). bit 1 (proper pairs):
	                       * the second slot's fragment is written first (alnfrags.c:1807-1812). Input of the stage-3c calls. */
} kmahip_hits;

/* Stage-2 result for paired reads (`-ipe ... -apm p`): two record slots per pair, in the order the
 * reference writes them to the S2 stream (printPair / deConPrintPtr, savekmers.c:3609-3737, ankers.c:150-160).
 * mate[r] = -1: slot not written; else 0/1 = which mate of the pair the record carries. rc[r] = 1: its
 * reverse complement is the sequence passed on. rc_flag / flag / T as in kmahip_cands; a properly paired
 * couple is "first record with an empty list (flag & 2), second record with the shared list". */
typedef struct kmahip_pe_recs {
	int32_t *mate;        /* 2 * n_pairs */
	int32_t *rc;
	int32_t *rc_flag;
	int32_t *flag;
	int64_t *R_off;       /* 2 * n_pairs + 1 */
	int32_t *T;
	int64_t T_cap;
} kmahip_pe_recs;

void kmahip_default_params(kmahip_params *p);
const char *kmahip_last_error(void);

/* device selection: one process per GPU (LOCAL_RANK), call before db_open */
int kmahip_init(int device);

int kmahip_db_open(const char *prefix, kmahip_db **out);
void kmahip_db_close(kmahip_db *db);
int kmahip_db_get_info(const kmahip_db *db, kmahip_db_info *info);

int kmahip_ws_create(kmahip_db *db, kmahip_ws **out);
void kmahip_ws_destroy(kmahip_ws *ws);

/* Stage 2, single-end `-1t1`.  Host buffers in, host buffers out (PCIe
 * inclusive).  Returns KMAHIP_EOVERFLOW if T_cap is too small; T_off[n] then
 * holds the needed capacity. */
int kmahip_scan_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                   const kmahip_params *p, kmahip_cands *out);
/* Same with everything resident in HBM; asynchronous on `stream`
 * (a hipStream_t, NULL = default stream). */
int kmahip_scan_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                       const kmahip_params *p, kmahip_cands *out, void *stream);
/* Stage 2, paired end with pairing penalty (`-apm p`): save_kmers_pair = save_kmers_penaltyPair
 * (savekmers.h:51, savekmers.c:3572-3777) with get_kmers_for_pair (:427-688). reads = mates interleaved
 * (read 2i = mate 1, 2i+1 = mate 2 of pair i). *_dev: device pointers, asynchronous on `stream`. */
int kmahip_scan_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, kmahip_pe_recs *out);
int kmahip_scan_pe_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                       kmahip_pe_recs *out, void *stream);

/* Stage 3a for the records of kmahip_scan_pe_dev (device pointers): alnFragsPE = alnFragsPenaltyPE
 * (alnfrags.h:68, alnfrags.c:1596-1972) for proper couples (first record with an empty list, second with the
 * shared candidates), alnFragsSE for records written singly; update_Scores_pe / update_Scores_se
 * (updatescores.c:300-488). The per-record arrays of `out` have 2 * n_pairs entries; pe_kind[pair] = 0 records
 * handled singly / nothing, 1 proper pair, 2 unmated pair, 3 first record only, 4 second record only. Hits of a
 * couple are stored in the SECOND record's slice [R_off[2p+1], ...): kind 1: n_hits shared by both records;
 * kind 2: the first record's n_hits[2p] hits followed by the second's n_hits[2p+1]; kind 3/4: that record's. */
int kmahip_align_pe_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                        const kmahip_params *p, kmahip_hits *out, int32_t *pe_kind, void *stream);
/* Stages 2 + 3a for paired reads with host buffers in and out. */
int kmahip_map_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                  kmahip_pe_recs *recs_out, kmahip_hits *hits_out, int32_t *pe_kind);

/* Stage 3a, single end: alignment score of every (read, candidate) pair,
 * per-read hit selection and ConClave accumulators.  `cands` is the output of
 * kmahip_scan_se_dev on the same `reads` (device pointers). */
int kmahip_align_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                        const kmahip_params *p, kmahip_hits *out, void *stream);
/* Stages 2 + 3a with host buffers in and out (reads staged once). */
int kmahip_map_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                  kmahip_cands *cands_out, kmahip_hits *hits_out);

/* Stage 3b, ConClave (runConClave, conclave.c:43-215, `-ConClave 1`): ONE template per frag_raw record, chosen from the
 * templates the read aligned equally well to by (alignment_scores, alignment_scores / template_length,
 * uniq_alignment_scores, smaller id). Inputs are the outputs of kmahip_align_se_dev / kmahip_align_pe_dev (device
 * pointers; with several GPUs AFTER kmahip_allreduce_scores). Per record slot (SE: read i; PE: record slot r, a proper
 * pair being the record of its second slot): the chosen signed template (0 = nothing written for the slot), its start
 * and end. Per template, ADDED into caller-zeroed vectors: w_scores (the Score column of `.res`, conclave.c:147),
 * fragment / read counts (:148-151, 172-174) and the summed read lengths (Depth under `-sasm`, assembly.c:1280).
 * With several GPUs these four are summed over ranks afterwards (kmahip_allreduce_scores takes any
 * two u64 vectors). */
typedef struct kmahip_conclave {
	int32_t *tmpl;               /* n slots */
	int32_t *start;
	int32_t *end;
	uint64_t *w_scores;          /* DB_size */
	uint32_t *fragment_counts;   /* DB_size, may be NULL */
	uint32_t *read_counts;       /* DB_size, may be NULL */
	uint64_t *depth;             /* DB_size, may be NULL */
} kmahip_conclave;
int kmahip_conclave_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                           const kmahip_hits *hits, kmahip_conclave *out, void *stream);
int kmahip_conclave_pe_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                           const kmahip_hits *hits, const int32_t *pe_kind, kmahip_conclave *out, void *stream);
/* the same with host buffers in and out (only reads->n_reads / len are used of `reads`; the per-template vectors of
 * `out` are read, added to and written back) */
int kmahip_conclave_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                       const kmahip_hits *hits, kmahip_conclave *out);
int kmahip_conclave_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                       const kmahip_hits *hits, const int32_t *pe_kind, kmahip_conclave *out);
/* The general form, host buffers: one entry per frag_raw record IN STREAM ORDER, exactly the fields runConClave reads
 * (conclave.c:59-70): hits->n_hits[r] listed templates at off[r] of tmpl / start / end, hits->best_score[r] = stats[2]
 * (negative: a pair record, the mate of length q_len2[r] follows), q_len[r]. off has n_records + 1 entries; q_len2 may be
 * NULL. For glue that merges single and paired results of one input stream: a record whose list is empty takes the first
 * listed hit of the record before it (the reference reads zero entries into buffers it does not clear). */
int kmahip_conclave_records(kmahip_db *db, kmahip_ws *ws, int64_t n_records, const int32_t *q_len, const int32_t *q_len2,
                            const int64_t *off, const kmahip_hits *hits, kmahip_conclave *out);
/* ... with DEVICE pointers, asynchronous on `stream`; `off` needs n_records entries only (a record with n_hits 0 and
 * score 0 is an unused slot: ConClave passes over it) */
int kmahip_conclave_records_dev(kmahip_db *db, kmahip_ws *ws, int64_t n_records, const int32_t *q_len, const int32_t *q_len2,
                                const int64_t *off, const kmahip_hits *hits, kmahip_conclave *out, void *stream);

/* The columns of a `.res` row that do not depend on the consensus (runkma.c:765-783, 809): Score, Expected (as printed,
 * (unsigned) expected), Template_length, q_value, p_value, and whether the template passes the reference's gate for
 * assembly / output (cmp_or(p <= evalue && score > expected, score >= scoreT * length), stdstat.c:23-27). One row per
 * template with w_scores > 0, in template order; w_scores is a HOST vector (summed over ranks). kma.c:312,319:
 * evalue = 0.05, scoreT = 0.5. Returns KMAHIP_EOVERFLOW with *n_rows = needed when cap is too small. */
typedef struct kmahip_res_row {
	int32_t template_id;
	int32_t template_length;
	uint64_t score;
	uint32_t expected;
	int32_t significant;
	double q_value;
	double p_value;
} kmahip_res_row;
/* How a template's p-value test and its score test combine into `significant` -- the reference's `cmp` pointer (stdstat.c:23-35,
 * kma.c:915-920): 0 = or (default), 1 = and (`-and`), 2 = always true (`-oa`, which also sets -ID and -md to 0). One setting per
 * process, read by kmahip_res_rows and by every run entry point (runkma.c:783, mt1.c:419). */
int kmahip_set_cmp(int mode);
/* `-lc` outside the chain finder: ConClavePtr = runConClave_lc (kma.c:694-701, conclave.c:215-385) -- among a read's equally good
 * templates the ConClave score per template base decides before the score itself. One setting per process; every ConClave entry
 * point reads it. (The chain finder's length-corrected helpers, kmeranker.c:37-55, 432-510, are not built: -lc needs -1t1.) */
int kmahip_set_conclave_lc(int on);
/* `-mem_mode` (runKMA_MEM instead of runKMA, kma.c:1619-1623, runkma.c:910-1250): ConClave is based on the template finder's own
 * scores -- a stage-2 record is the frag_raw record (its templates as the hits, each spanning its template, the k-mer score as the
 * read score; update_Scores_MEM updatescores.c:31-67), no alignment happens before ConClave; stage 3c aligns every read against the
 * template ConClave gave it as ever. A couple of a paired stream -- a first record without a list, then its mate with the templates --
 * is one record with both scores added up (update_Scores_pe_MEM :69-107, runkma.c:1090-1134). One setting per process, read by the
 * whole-run entry points (kmahip_run_se / _pe / _chain, the sessions, the sharded runs). */
int kmahip_set_mem_mode(int on);
int kmahip_res_rows(const kmahip_db *db, const uint64_t *w_scores, double evalue, double scoreT,
                    kmahip_res_row *rows, int64_t cap, int64_t *n_rows);

/* Stage 3c, per read: the traceback aligner KMA() (align.c:214-507 with NW / NW_band, nw.c:26-640) for every read that
 * ConClave filed under a template, followed by assemble_KMA's read filter (assembly.c:1931-1961: + Wl for an alignment
 * that starts at the first / ends at the last template base, minlen, mrc, scoreT). This is what the reference prints per
 * read in `.frag.gz` / SAM and feeds to alnToMat.
 *   rc[i]     kmahip_hits.rc of the read's record (bit 0: the filed fragment is the reverse complement of reads[i])
 *   tmpl[i]   kmahip_conclave.tmpl (signed; 0 = read has no template)
 *   tmpl_ok   per template, 1 = assemble (kmahip_res_row.significant); NULL = all
 * Out per read: stats[10 * i ..] = score, start, end, aln_len, clip_start, clip_end, match, tGaps, qGaps, mapQ (all 0: the
 * read was dropped); the alignment columns as n_ops[i] runs at ops[ops_off[i] ..], each (length << 2) | class with class
 * 0 '=' (match), 1 'X' (mismatch), 2 'I' (gap in template), 3 'D' (gap in read) -- the classes of makeCigar (sam.c:57-78).
 * The SAM CIGAR is clip_start 'S' + runs + clip_end 'S'; POS = start + 1; AS = score. Runs are appended in no particular
 * read order. KMAHIP_EOVERFLOW: ops_cap too small (kmahip_ws_status). Device pointers; asynchronous on `stream`. */
typedef struct kmahip_traces {
	int32_t *stats;     /* 10 * n_reads */
	int64_t *ops_off;   /* n_reads */
	int32_t *n_ops;     /* n_reads */
	uint32_t *ops;      /* ops_cap */
	int64_t ops_cap;
} kmahip_traces;
int kmahip_align_trace_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                           const uint8_t *tmpl_ok, const kmahip_params *p, kmahip_traces *out, void *stream);
/* the same with host buffers in and out; returns KMAHIP_EOVERFLOW with the needed run count in *ops_needed */
int kmahip_align_trace(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                       const uint8_t *tmpl_ok, const kmahip_params *p, kmahip_traces *out, int64_t *ops_needed);

/* `-Mt1 t` (runKMA_Mt1, mt1.c:86-500 -> assemble_KMA with read_score == 0, assembly.c:1917-1965): raw reads go straight to
 * stage 3c against ONE template, no stage 2 and no ConClave. Per read: anker_rc (align.c:780-991) seeds both strands against
 * the template's position index and keeps the strand with the larger MEM coverage (forward on equality; the forward strand is
 * only seeded when one of its every-k-th k-mers is in the index, preseed align.c:750-768, unless p->exhaustive), KMA() chains
 * its MEMs and joins them with traceback, then the read filter as in kmahip_align_trace_dev. one2one: `-1t1` was given too
 * (anker_rc's coverage gate, align.c:952). Out: kmahip_traces as above, and rc_out[i] (may be NULL) = 1 when the reverse
 * complement of read i is what was aligned -- pass it as `rc` and a vector of `tmpl` to kmahip_assemble2. Built as a pipeline
 * of kernels (seed + chain per read on a wavefront, DP problems batched by size); device pointers. */
int kmahip_align_trace_mt1_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, int32_t tmpl, int one2one, const kmahip_params *p,
                               kmahip_traces *out, int32_t *rc_out, void *stream);
/* the same with host buffers in and out; KMAHIP_EOVERFLOW with the needed run count in *ops_needed */
int kmahip_align_trace_mt1(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, int32_t tmpl, int one2one, const kmahip_params *p,
                           kmahip_traces *out, int32_t *rc_out, int64_t *ops_needed);

/* Stage 3c per template: pile-up of the traced reads (alnToMat, assembly.c:1317-1444: per template position the counts of
 * A C G T N and gap, insertion columns chained between positions) on the device, then callConsensus + baseCaller
 * (assembly.c:1499-1631, 162-179) in host arithmetic. Inputs: the reads, the rc / tmpl arrays given to
 * kmahip_align_trace and its output (all HOST buffers here). Reads of one template are piled up in the reference's order:
 * reverse stream order inside every chunk of max_frag filed fragments (conclave.c:164-166, 194; kma.c default 1000000;
 * <= 0 selects it) -- only the gap count a NEW insertion column starts with depends on it. bcd / evalue: `-bcd` (1) and
 * `-e` (0.05). Out, per template (DB_size entries, zero where nothing was piled up): cover = called positions equal to the
 * template base, aln_len = called columns, depth = summed depth of the called columns, asm_len = columns incl.
 * insertion columns (runkma.c:792-800 turns these into Template_Identity / Template_Coverage / Query_Identity /
 * Query_Coverage / Depth). Optionally the consensus line of every assembled template ("ACGTN-", lower case = call not
 * significant, '-' = no call), 0-terminated, at consensus + consensus_off[t]. */
typedef struct kmahip_assembly {
	int64_t *cover;
	int64_t *aln_len;
	int64_t *depth;
	int64_t *asm_len;
	char *consensus;          /* may be NULL */
	int64_t *consensus_off;   /* DB_size, may be NULL */
	int64_t consensus_cap;
	int64_t consensus_used;   /* in: 0; out: bytes written */
} kmahip_assembly;
int kmahip_assemble(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                    const kmahip_traces *traces, int64_t max_frag, int bcd, double evalue, kmahip_assembly *out);
/* kmahip_assemble with the per-read inputs already in HBM (reads, rc, tmpl and the traces are DEVICE pointers, e.g. the
 * outputs of kmahip_align_trace_dev); `out` is filled on the host as above. */
int kmahip_assemble_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                        const kmahip_traces *traces, int64_t max_frag, int bcd, double evalue, kmahip_assembly *out);

/* kmahip_assemble with the knobs of the other stage-3c flavours: order 1 = pile the reads up in stream order (what the single
 * reading thread of `-Mt1` does, assembly.c:1873-1965) instead of ConClave's per-template order; caller 1 = nanoCaller
 * (assembly.c:205-240) and sig90 1 = significantAnd90Nuc (assembly.c:147-149), the pair `-bcNano` selects (kma.c:762-766). */
typedef struct kmahip_assemble_opts {
	int64_t max_frag;   /* <= 0: 1000000 */
	double evalue;      /* -e, 0.05 */
	int32_t bcd;        /* -bcd, 1 */
	int32_t order;      /* 0 ConClave's order, 1 stream order */
	int32_t caller;     /* 0 baseCaller, 1 nanoCaller (-bcNano), 2 orgBaseCaller (-bcg), 3 refCaller (-ref_fsa), 4 refNanoCaller (-ref_fsa
	                     * -bcNano): assembly.c:162-270. + 8: insertion columns called as gaps -- which the reference trims from its alignment,
	                     * assembly.c:748-752 -- come out as '_' in the consensus string instead of '-' (for a writer that keeps the gaps of
	                     * template positions: -ref_fsa 0). + 16: alnToMatDense (-dense, assembly.c:1446-1497): template positions only, no
	                     * insertion columns. + 32: the character of EVERY insertion column carries bit 7 (what kmahip_aln_entry needs to tell
	                     * them from template positions; a reader of the string masks with 0x7F) */
	int32_t sig90;      /* 0 significantNuc, 1 significantAnd90Nuc (-bc90, -bcNano), 2 significantAndSupport (-bc x): `support` below */
	/* per read: how many filed fragments (ConClave template != 0) precede it in the WHOLE stream -- what the reference's
	 * chunks of max_frag records are counted in (conclave.c:166, 194). NULL: the batch is the whole stream and the
	 * positions are counted here. A rank that piles up the reads of its templates gathered from several read shards
	 * (kma_amd/dist.py) passes the positions the reads had in the global stream. Host pointer for kmahip_assemble2,
	 * device pointer for kmahip_assemble2_dev. */
	const int64_t *frag_rank;
	double support;     /* sig90 == 2: a base call needs support * depth of the column (assembly.c:151-160) */
} kmahip_assemble_opts;
int kmahip_assemble2(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                     const kmahip_traces *traces, const kmahip_assemble_opts *opts, kmahip_assembly *out);
int kmahip_assemble2_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                         const kmahip_traces *traces, const kmahip_assemble_opts *opts, kmahip_assembly *out);

/* The whole single-end `-1t1` run on one batch, one call: reads uploaded ONCE, stage 2, stage 3a, ConClave, the `.res`
 * statistics, the traceback aligner and the pile-up all on what is already in HBM; only the per-template results (and, if
 * asked for, the per-read columns a `.frag.gz` / SAM writer needs) come back. Equivalent to kmahip_map_se +
 * kmahip_conclave_se + kmahip_res_rows + kmahip_align_trace + kmahip_assemble, minus four uploads of the reads and the
 * host round trips in between (runKMA, runkma.c:104-900, for one chunk of input). HOST buffers.
 *   rows / rows_cap / n_rows   as kmahip_res_rows (KMAHIP_EOVERFLOW with n_rows = needed)
 *   assembly                   as kmahip_assemble
 *   tmpl, n_hits, rc, trace_stats   optional per-read outputs (n_reads, n_reads, n_reads, 10 * n_reads), NULL to skip
 *   ms[6]                      wall time of: upload, stages 2 + 3a, ConClave + statistics, traceback, pile-up, consensus (host)
 *   caller, sig90              IN: nanoCaller / significantAnd90Nuc for the consensus (`-bcNano`, kma.c:762-766); zero = baseCaller */
typedef struct kmahip_run {
	kmahip_res_row *rows;
	int64_t rows_cap, n_rows;
	kmahip_assembly assembly;
	int32_t *tmpl, *n_hits, *rc, *trace_stats;
	double ms[6];
	int32_t caller, sig90;      /* IN, kmahip_run_se / _pe / _chain: the base caller as in kmahip_assemble_opts (`-bcNano` = 1, 1; 0, 0 = baseCaller) */
	double support;             /* IN: as kmahip_assemble_opts.support */
} kmahip_run;
int kmahip_run_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, double evalue, int bcd,
                  int64_t max_frag, kmahip_run *out);

/* The `-Mt1 tmpl` run on one batch (runKMA_Mt1, mt1.c:86-500): raw reads uploaded once, kmahip_align_trace_mt1 per read, the
 * pile-up in stream order and the consensus, all on the device. HOST buffers. aopts: evalue, bcd, caller / sig90 (`-bcNano` = 1, 1);
 * order is stream order whatever aopts says. Out: rows[0] (n_rows = 1) = the `.res` figures of mt1.c:425-447: Score = summed KMA()
 * scores of the kept reads, Expected 0, q_value = Score, p_value = p_chisqr(Score), significant = the reference's gate for printing
 * the row with its identity columns; assembly as kmahip_assemble; tmpl[i] = tmpl for the reads that were kept, n_hits[i] = 1,
 * rc[i] = strand, trace_stats as kmahip_traces.stats -- what kmahip_frag_write2 (order 1) takes. */
int kmahip_run_mt1(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, int32_t tmpl, int one2one, const kmahip_params *p,
                   const kmahip_assemble_opts *aopts, kmahip_run *out);

/* One `.res` row exactly as runKMA prints it (runkma.c:809) from kmahip_res_rows + kmahip_assemble; returns the number of
 * characters written, 0 when the reference prints no row for the template (nothing covered, identity below -ID (1.0) or
 * depth below -md (0.0)). */
int kmahip_res_line(const char *template_name, const kmahip_res_row *row, int64_t cover, int64_t aln_len, int64_t depth_sum,
                    double ID_t, double Depth_t, char *line, int64_t cap);

/* ---- stage 2 of the DEFAULT mode (no -1t1) -------------------------------------------------------------------
 * kmerScan = save_kmers_chain (savekmers.c:5127-5945, the reference's default, savekmers.c:40) with the default helpers of
 * kmeranker.c:25-30. A read yields zero or more S2 records, one per accepted chain of anchors: rc_flag = the chain's score
 * (negative: both strands carry it and the reverse strand's templates follow as negative ids), emit_rc = 1 when the record
 * prints the reverse-complemented read (print_ankers with qseq_r), [q_start, q_end) = the query bounds insertKmerBound appends
 * to the header behind a NUL (qseqs.c:41-56; read back at alnfrags.c:1092-1099, conclave.c:137-145, assembly.c:1926-1933).
 * Records come back in stream order: reads ascending, the chains of a read in the order the reference prints them. HOST buffers;
 * KMAHIP_EOVERFLOW with n_recs / n_T = what is needed when rec_cap / T_cap are too small. Parameters: minlen (-ml, 16),
 * coverT (-mct, 0.1: how much of a chain may overlap what was taken before), mrs (-mrs, 0.5); NULL = those defaults.
 * One lane per read on per-lane scratch in HBM: the anchor search is this round's correct-first form. */
typedef struct kmahip_chain_params { int32_t minlen; int32_t pad_; double coverT, mrs; } kmahip_chain_params;
typedef struct kmahip_chain_recs {
	int64_t rec_cap, T_cap;   /* in: capacities of the arrays below (T_off: rec_cap + 1) */
	int64_t n_recs, n_T;      /* out */
	int64_t *read;            /* index of the read in the batch */
	int32_t *rc_flag, *emit_rc, *q_start, *q_end;
	int64_t *T_off;           /* templates of record i: T[T_off[i] .. T_off[i + 1]) */
	int32_t *T;
} kmahip_chain_recs;
int kmahip_scan_chain(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, const kmahip_chain_params *cp,
                      kmahip_chain_recs *out);
/* One place where the reference's own result is not a function of its input (savekmers.c:5447-5449): behind an N the chain finder
 * restarts the reverse strand's rolling k-mer k bases too far on, which for a read with an N among its first k - 1 bases lies
 * beyond the end of that read in a buffer the reference never clears -- zeros while no longer read came before it in the stream,
 * otherwise what the longer read left there (and with -t > 1 whichever read the thread's buffer held last). This library reads
 * zeros. kmahip_chain_unpinned_reads counts, on HOST arrays of a batch in stream order, the reads for which that matters: an N among
 * the first k - 1 bases and a longer read somewhere before (longest_before: the longest read of the batches before this one, 0 for
 * the first; *longest_after = what to pass with the next batch). Their records may differ from the reference's; a host that must
 * know says so (examples/kmahip_map prints the count). */
int kmahip_chain_unpinned_reads(const int32_t *len, const int32_t *N, const int64_t *N_off, int64_t n_reads, int k,
                                int32_t longest_before, int32_t *longest_after, int64_t *count);

/* ---- index build: `kma index -i <fasta ...> -o <prefix> [-k k]` ------------------------------------------------
 * Writes <prefix>.comp.b / .length.b / .seq.b / .name as the reference's index.c + makeindex.c:167-330 (makeDB) +
 * compress.c:83-614 (compressKMA_DB) do for the default options: the hashed index form, k <= 16, templates trimmed of leading /
 * trailing N's (their count goes into the name as " B<n>"), templates shorter than k skipped, k-mers that hold an N left out,
 * equal template lists stored once. Every k-mer start becomes a 64-bit key on the device (k-mer << 32 | template), sorted and
 * made unique there (rocPRIM); grouping, list sharing and the bucket directory are one pass on the host. The files hold the
 * same k-mer -> template-list mapping as the reference's (tests compare the two) and the reference maps against them with
 * identical results; bucket count, key order inside a bucket and list order are the builder's own. FASTA input, plain or .gz.
 * Not covered: -Sparse / prefixes, minimizers (-m), homopolymer compression (-hc), -deCon, appending (-t_db); `-batch list` is the
 * host program's (examples/kmahip_index: the listed paths become fasta_paths). */
int kmahip_index_build(const char *const *fasta_paths, int n_files, const char *out_prefix, int kmersize);

/* ---- stage 1: FASTQ / FASTA ingest into packed read batches -------------------------------------------
 * Host code (zlib for .gz, plain files read as they are): the record parser of FileBuffgetFq / FileBuffgetFsa
 * (seqparse.c:241-403, 66-159) with the to2Bit table of kma.c:1440-1480 (IUPAC codes fold onto ACGT, N / X = 4), the
 * phred-scale guess of getPhredFileBuff (seqparse.c:551-589), the quality trimming of phredStat (runinput.c:127-313;
 * fsastat :315-368 for FASTA), the length gate of run_input / run_input_PE (runinput.c:370-606) and the 2-bit packing of
 * compDNA (compdna.c:99-127). What it yields is what the reference's stage 1 writes into the S1 stream (printFsa /
 * printFsa_pair, runinput.c:765-830), in the same order. */
typedef struct kmahip_trim {
	int32_t min_phred;    /* -mp  (20): 5' / 3' end bases below it are trimmed */
	int32_t min_q;        /* -eq  (0): minimum average quality, bi-directional trimming towards it */
	int32_t hardmask_q;   /* -mi  (0): bases below it become N */
	int32_t min_len;      /* -ml  (16) */
	int32_t max_len;      /* -xl  (2147483647) */
} kmahip_trim;
void kmahip_trim_default(kmahip_trim *t);

typedef struct kmahip_ingest kmahip_ingest;

/* One batch, owned by the reader and valid until the next call on it. `reads` holds HOST pointers in the layout every
 * kmahip_*_se / _pe call takes (each read followed by one pad word). pair[i]: 0 = single record, 1 = first mate of a
 * pair record (its second mate is read i + 1, pair 2). */
typedef struct kmahip_read_batch {
	kmahip_reads reads;
	const char *names;        /* header lines without '@' / '>', chomped, each NUL-terminated */
	const int64_t *name_off;  /* n_reads + 1 */
	const uint8_t *pair;      /* n_reads */
	int64_t records;          /* S1 records in the batch (a pair counts once) */
} kmahip_read_batch;

/* path2 == NULL: single end (run_input); otherwise the two mate files are read in lockstep (run_input_PE): both mates
 * long enough -> a pair record, one of them -> a single record, none -> dropped. */
int kmahip_ingest_open(const char *path1, const char *path2, const kmahip_trim *trim, kmahip_ingest **out);
/* `-int file` (run_input_INT, runinput.c:608-740): ONE file whose records are taken two at a time as the mates of a couple, trimmed
 * and gated like the two files of -ipe (both long enough -> a pair record, one -> a single record); the last record of a file with
 * an odd number of them meets an empty mate (FileBuffgetFq clears the length before it finds the end, seqparse.c:249) and is filed
 * singly. The batches are those of a paired reader (pair[] = 1 / 2 / 0). FASTQ only: for FASTA the reference cuts mate 2 with
 * mate 1's bounds (runinput.c:705), which may reach past what it read -- KMAHIP_EFORMAT. */
int kmahip_ingest_open_interleaved(const char *path, const kmahip_trim *trim, kmahip_ingest **out);
/* up to max_records further S1 records; batch->reads.n_reads == 0 at the end of the input. A record that does not start
 * with '@' ends the input like in the reference ("Malformed input.", seqparse.c:256-260): the records before it are
 * delivered, then one call returns KMAHIP_EFORMAT. */
int kmahip_ingest_next(kmahip_ingest *in, int64_t max_records, kmahip_read_batch *batch);
/* A second bound on the batches of kmahip_ingest_next, for inputs of long reads where a count of records says little about a batch's
 * size: a batch also closes once it holds about max_bases bases (checked between the stretches of input the reader cuts into records:
 * it may go over by one such stretch, tens of megabytes of text). 0 (the default): no such bound. */
int kmahip_ingest_set_batch_bases(kmahip_ingest *in, int64_t max_bases);
/* For a caller that took the whole input as one batch (max_records = INT64_MAX): KMAHIP_EIO / KMAHIP_EFORMAT when the input broke
 * off behind the records delivered (a truncated or corrupt .gz, a record that does not start with '@'; the reference ends with a
 * non-zero exit status there), 0 otherwise. Does not touch the batch -- another kmahip_ingest_next would, its arrays are reused. */
int kmahip_ingest_status(kmahip_ingest *in);
/* 33 or 64 (0: undeterminable, treated like the reference does); records read / kept so far */
int kmahip_ingest_phred_scale(const kmahip_ingest *in);
void kmahip_ingest_counts(const kmahip_ingest *in, int64_t *records_read, int64_t *records_kept);
void kmahip_ingest_close(kmahip_ingest *in);

/* `.frag.gz` (updateFrags, assembly.c:49-83): one row per read that passed the stage-3c filter -- the read as aligned, the
 * number of equally good templates, score, start, end, template name (from <prefix>.name), read header -- in the order a
 * single-threaded assemble_KMA writes them: templates ascending, inside a template the pile-up order (see kmahip_assemble).
 * HOST buffers: the reads, kmahip_hits.rc, kmahip_conclave.tmpl, the per-read tie count (kmahip_hits.n_hits), the stats
 * array of kmahip_align_trace and the read headers (kmahip_read_batch.names / name_off). A path ending in ".gz" is
 * gzip-compressed (level 1 as filebuff.c:189), anything else plain text. */
int kmahip_frag_write(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                      const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, const char *read_names,
                      const int64_t *read_name_off, int64_t *rows);

/* kmahip_frag_write with the row order chosen: order 0 = as above, 1 = stream order inside a template (`-Mt1`) */
int kmahip_frag_write2(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                       const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, int order, const char *read_names,
                       const int64_t *read_name_off, int64_t *rows);

/* kmahip_frag_write2 for a batch that is not the whole stream: frag_rank[i] = number of filed fragments before read i in
 * the whole stream (as kmahip_assemble_opts.frag_rank; NULL = count them in this batch). */
int kmahip_frag_write3(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                       const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, int order, const int64_t *frag_rank,
                       const char *read_names, const int64_t *read_name_off, int64_t *rows);

/* One gzip member as kmahip_frag_write* makes them for a path ending in ".gz": Huffman coding only (the reference deflates
 * at level 1, filebuff.c:189 -- what a reader inflates is the same). The writers compress blocks of rows on several threads
 * and concatenate the members (RFC 1952 2.2). Exposed for tests. */
int kmahip_gzip_member(const void *src, int64_t n, void *dst, int64_t cap, int64_t *out_bytes);

/* The single-end run in KMA's DEFAULT mode (no -1t1) on one batch: kmahip_scan_chain, then every S2 record -- a read, or its
 * reverse complement where the record prints that, with its query bounds -- goes through stage 3a, ConClave, the `.res`
 * statistics, the traceback and the pile-up like a read of kmahip_run_se (runKMA, runkma.c:104-900 with kmerScan =
 * save_kmers_chain). A chimeric read therefore counts once per chain. `names` / `name_off`: read headers for the `.frag(.gz)`
 * rows written to frag_path (both NULL to skip the file). out->tmpl / n_hits / rc / trace_stats are not filled (the records
 * are not the caller's reads); rows, assembly and ms are as in kmahip_run_se. */
int kmahip_run_chain(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const char *names, const int64_t *name_off,
                     const kmahip_params *p, const kmahip_chain_params *cp, double evalue, int bcd, int64_t max_frag,
                     const char *frag_path, kmahip_run *out);

/* The paired run (`-ipe r1 r2 -apm p -1t1`) on one batch as kmahip_ingest_next hands it over for two mate files: reads in
 * stream order, batch->pair[i] = 1 / 2 for the mates of a pair record, 0 for a record that lost its mate to the trimming.
 * Pairs go through kmahip_map_pe, single records through kmahip_map_se; their results are merged into frag_raw records in
 * stream order (alnFragsPenaltyPE's record forms: proper pair -- second slot first when the pair was swapped --, unmated,
 * one mate only, alnfrags.c:1777-1970), ConClave over the records, the `.res` statistics, the traceback aligner over the
 * fragments in record order and the pile-up. Fills out->rows / n_rows / assembly / ms (the per-read pointers of `out` are
 * not used: fragments are not in read order); frag_path != NULL also writes the `.frag.gz`. HOST buffers. */
int kmahip_run_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_read_batch *batch, const kmahip_params *p, double evalue, int bcd,
                  int64_t max_frag, const char *frag_path, kmahip_run *out);

/* Paired input in KMA's DEFAULT mode (`-ipe r1 r2` without -1t1): the reference's batch loop hands a couple to save_kmers_pair
 * and a record that lost its mate to kmerScan (save_kmers_batch, savekmers.c:171-200), which without -1t1 is save_kmers_chain
 * (savekmers.c:40) -- the couples are mapped exactly as under -1t1 (one2one only guards a branch of anker_rc / anker_rc_comp no
 * seeded read reaches, align.c:964,1157), a single record yields the chain finder's records with their query bounds. With `cp`
 * set on the workspace every paired run on it (kmahip_run_pe, kmahip_run_pe_sharded, a session after kmahip_session_set_pe)
 * does that: each record of a single read is a unit of the stream where its read stood, and goes through stage 3a, ConClave,
 * the traceback and the `.frag.gz` like a record of kmahip_run_chain. NULL: back to -1t1 (single records through save_kmers). */
int kmahip_ws_set_pe_chain(kmahip_ws *ws, const kmahip_chain_params *cp);

/* Multi-GPU (one process per GPU): in-place SUM over all ranks of the two ConClave
 * vectors on `stream`, through RCCL (ncclAllReduce, ncclUint64, ncclSum).
 * `nccl_comm` is an ncclComm_t the host program created (ncclCommInitRank);
 * librccl is resolved at run time, the library does not link against it. */
int kmahip_allreduce_scores(void *nccl_comm, uint64_t *alignment_scores, uint64_t *uniq_alignment_scores,
                            size_t DB_size, void *stream);

/* ---- several GPUs of one node: one process per GPU, the reads sharded over the ranks -----------------------------
 * The reference is one process (threads over pipes, kmapipe.c:55-146); these entries are what a read-sharded host needs between
 * its stages: the SUM of alignment_scores / uniq_alignment_scores before runConClave (runkma.c:563-594, updatescores.c:228,276),
 * the SUM of ConClave's per-template outputs before the `.res` statistics (conclave.c:147-151, runkma.c:608-613, 770-783), and the
 * exchange that brings every traced read to the rank that owns its template, in the order of the whole stream (the assembly
 * order of conclave.c:164-196 and assembly.c:1377-1424 is made of it).
 * A communicator is bootstrapped through a POSIX shared-memory segment named after `key` (every rank of one run passes the same
 * key, no two runs at a time the same one); backend "rccl": device data moves with ncclAllReduce / ncclSend / ncclRecv over xGMI
 * (one device per rank), "shm": staged through host memory (any number of ranks per device: rehearsals and tests). */
typedef struct kmahip_comm kmahip_comm;
int kmahip_comm_init(int rank, int world, const char *key, const char *backend, kmahip_comm **out);
void kmahip_comm_destroy(kmahip_comm *c);
int kmahip_comm_rank(const kmahip_comm *c);
int kmahip_comm_world(const kmahip_comm *c);
int kmahip_comm_is_rccl(const kmahip_comm *c);
/* One line about the communicator for a log or a JSON record: "backend=rccl rank=R world=W rccl_nranks=N rccl_rank=R
 * rccl_version=V allreduces=A alltoallvs=B" -- nranks / rank / version as RCCL itself reports them (ncclCommCount,
 * ncclCommUserRank, ncclGetVersion), A and B the exchanges that went through it so far --, or "backend=shm|none rank=R world=W".
 * Returns what snprintf returns. With KMAHIP_COMM_FORCE_RCCL=1 a ONE-rank communicator of backend "rccl" is a real RCCL
 * communicator too (kmahip_comm_init runs a SUM all-reduce and a grouped send / recv through it before it returns, as it does
 * for every RCCL communicator): the transport can be exercised on a box with a single device. */
int kmahip_comm_describe(const kmahip_comm *c, char *buf, size_t cap);
int kmahip_comm_barrier(kmahip_comm *c);
/* every rank posts `bytes` (at most 64 KiB) of HOST memory and gets all ranks' back in rank order */
int kmahip_comm_allgather(kmahip_comm *c, const void *mine, size_t bytes, void *all);
/* in-place SUM over the ranks of n u64 values in DEVICE memory */
int kmahip_comm_allreduce_u64(kmahip_comm *c, uint64_t *d_buf, size_t n, void *stream);
/* all-to-all of byte blocks: send_bytes[d] bytes for rank d back to back in `send`; recv_bytes[s] from rank s land back to back
 * in `recv` in rank order (the sizes are agreed on beforehand, e.g. through kmahip_comm_allgather). device != 0: DEVICE buffers. */
int kmahip_comm_alltoallv(kmahip_comm *c, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes,
                          int device, void *stream);

/* Stage 1 for one rank of a sharded run: the part `part` of `parts` of the input. FASTQ in a plain file (single end): the rank
 * parses only its byte range -- ranges are cut at record starts found by the reader's own guess (a line beginning with '@' whose
 * next line but one begins with '+'), the same on every rank, so the parts tile the file; *whole_input = 0. Anything else (.gz,
 * FASTA, two mate files): the reader delivers the whole input and *whole_input = 1 -- the caller keeps the records
 * [n part / parts, n (part + 1) / parts) of it. path2 == "" (the empty string) asks for the interleaved reader of
 * kmahip_ingest_open_interleaved on path1. */
int kmahip_ingest_open_part(const char *path1, const char *path2, const kmahip_trim *trim, int part, int parts, kmahip_ingest **out,
                            int *whole_input);

/* The single-end `-1t1` run of kmahip_run_se + the three writers, with the reads sharded over the ranks of `comm` (NULL or a
 * one-rank communicator: everything on this device). `batch`: this rank's contiguous part of the input stream, in stream order
 * over the ranks. Each rank maps its reads (stages 2, 3a), the score vectors are summed, ConClave runs per shard on the global
 * vectors, its per-template outputs are summed (every rank then computes the same `.res` statistics), the traceback runs on the
 * rank's own reads, and every kept read travels to the owner of its template (contiguous template ranges, balanced by filed
 * fragments) with its position among the filed fragments of the whole stream; the owners pile up, call the consensus and write
 * the rows of their templates to <out_prefix>.part<rank>.res, .fsa and .frag.gz. Rank 0 then concatenates the parts in
 * rank order (= template order; gzip members concatenate) into <out_prefix>.res / .fsa / .frag.gz -- byte for byte the files of
 * the one-device run (tests/test_shard_gpu.py). ms[8]: upload, stages 2 + 3a, exchange 1 + ConClave + exchange 2, traceback,
 * gather by owner, pile-up + consensus, writers, merge. */
typedef struct kmahip_shard_opts {
	double evalue;        /* -e, 0.05 */
	int32_t bcd;          /* -bcd, 1 */
	int32_t caller, sig90;/* -bcNano: 1, 1 */
	int64_t max_frag;     /* -mf, <= 0: 1000000 */
	double ID_t, Depth_t; /* -ID (1.0), -md (0.0) */
	double support;       /* -bc x with sig90 == 2 */
	int32_t ref_fsa;      /* the consensus file: 0 gap columns left out, 1 gaps as n (-ref_fsa), 2 as they are (-ref_fsa 0; printconsensus.c:38-60) */
	int32_t write_aln;    /* != 0: <prefix>.aln as well (printConsensus printconsensus.c:26-37; the reference writes it unless -na) */
} kmahip_shard_opts;
/* One template's block of the `.aln` file: "# name", then per 60 alignment columns the lines "template:", the match line ('|' where
 * the call equals the template's base, '_' elsewhere) and "query:", of the alignment as assemble_KMA trims it (assembly.c:2094-2119:
 * without the insertion columns called as gaps). cons: the template's consensus string from kmahip_assemble2 / kmahip_run_* with
 * caller + 32. Returns the number of bytes written to out (no terminator), -1 on an error (cap too small: (columns / 60 + 2) * 224 +
 * strlen(name) + 16 always suffices). Host memory; reads <prefix>.seq.b the first time. */
int64_t kmahip_aln_entry(kmahip_db *db, int32_t tmpl, const char *name, const char *cons, char *out, int64_t cap);
int kmahip_run_se_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, const kmahip_params *p,
                          const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]);
/* The default mode (no -1t1; kmahip_run_chain) over read shards: stage 2 -- save_kmers_chain -- on the rank's reads, its records (a
 * read or its pieces with their query bounds) in stream order, and from there the stages and exchanges of kmahip_run_se_sharded on
 * records instead of reads; a fragment row carries the header of the read its record came from. cp: NULL = the defaults. */
int kmahip_run_chain_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, const kmahip_params *p,
                             const kmahip_chain_params *cp, const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]);
/* `-Mt1 tmpl [-bcNano]` (kmahip_run_mt1; runKMA_Mt1 mt1.c:86-500) over read shards: every rank traces its contiguous part of the stream
 * against the one template (the traceback is four fifths of that run), the Score and the number of kept reads are summed, and the kept
 * reads travel to the template's owner (rank 0) with their positions among the kept reads of the whole stream, where they are piled
 * up in stream order. Files as kmahip_run_se_sharded leaves them, identical to the one-process run. */
int kmahip_run_mt1_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, int32_t tmpl, int one2one,
                           const kmahip_params *p, const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]);
/* The paired run (`-ipe r1 r2 -apm p -1t1`, kmahip_run_pe) the same way. `batch`: this rank's contiguous part of the stream of
 * units (pairs and single records; a pair is never cut). On top of the three exchanges above, two things cross the shards because
 * runConClave walks ONE stream: a record whose hit list came out empty takes the first listed hit of the last record before it
 * that had one (conclave.c:123-127) -- the shards hand that hit on --, and the chunks of maxFrag filed fragments close one after
 * the other (conclave.c:164-196) -- the ranks count theirs in turn. */
int kmahip_run_pe_sharded(kmahip_db *db, kmahip_ws *ws, kmahip_comm *comm, const kmahip_read_batch *batch, const kmahip_params *p,
                          const kmahip_shard_opts *opts, const char *out_prefix, double ms[8]);

/* ---- the single-end `-1t1` run fed batch by batch: the host holds one batch at a time ------------------------------------------
 * The reference streams its input through pipes and spills the frag_raw records of stage 3a to a temporary file that ConClave
 * and the assembly read back once the input has ended (kmapipe.c:55-146, updatescores.c:283-295, runkma.c:563-594, 757-863). Here
 * HBM is that temporary file: kmahip_session_add uploads a batch of kmahip_ingest_next (reads, N positions, headers) behind the
 * batches before it, runs stages 2 and 3a on it and adds into the ConClave vectors -- the batch's host arrays may be reused when it
 * returns; kmahip_session_finish runs ConClave and the traceback per batch, one pile-up + consensus over everything, and writes
 * <out_prefix>.res, .fsa (write_fsa) and .frag.gz (write_frag) -- the fragment rows are ordered, measured and formatted on the
 * device and come back as text a chunk at a time for the host's threads to compress. Files byte for byte those of kmahip_run_se +
 * kmahip_frag_write (the .gz after inflating). opts: evalue, bcd, caller / sig90, max_frag, ID_t, Depth_t as in
 * kmahip_run_se_sharded; reads_hint: an estimate of the number of reads (0: none), sizes the device arrays up front.
 * ms[8]: uploads and stages 2 + 3a (each summed over the batches), ConClave + statistics, traceback, pile-up + consensus,
 * .res + .fsa, fragment rows. */
typedef struct kmahip_session kmahip_session;
int kmahip_session_open(kmahip_db *db, kmahip_ws *ws, const kmahip_params *p, const kmahip_shard_opts *opts, int64_t reads_hint, kmahip_session **out);
/* Paired input (`-ipe r1 r2 -apm p -1t1`; kmahip_run_pe) through the session: the batches of a paired reader (kmahip_ingest_open with
 * two files: batch->pair says which reads are mates; a batch never ends inside a couple) are uploaded behind each other like single-end
 * ones -- the host holds one batch at a time --, the first one is run through the stages once to pay for first launches and scratch
 * while stage 1 reads on, and kmahip_session_finish runs kmahip_run_pe's stages on everything and writes the files. Before the first
 * batch. */
int kmahip_session_set_pe(kmahip_session *s);
/* The reference's DEFAULT mode (no -1t1: kmahip_run_chain) through the same session: call once, before the first batch. A batch's
 * reads then go through the chain finder, and what the session keeps and maps are its records (a read, or its pieces, with their query
 * bounds); a fragment row carries the header of the read its record came from. cp: NULL = the defaults (kmahip_scan_chain). */
int kmahip_session_set_chain(kmahip_session *s, const kmahip_chain_params *cp);
/* `-Mt1 tmpl` (kmahip_run_mt1; runKMA_Mt1, mt1.c:86-500) through the same session: call once, before the first batch. There is no
 * stage 2 and no ConClave in this mode, and a read's traceback depends on nothing but the read: kmahip_session_map seeds and traces
 * the batch's reads against the template right away -- beside stage 1 of the next batch -- and keeps their figures and alignment runs
 * in HBM; kmahip_session_finish sums the `.res` row, piles everything up in stream order and writes the three files (those of
 * kmahip_run_mt1 + kmahip_frag_write2 with order 1). frag_path (or NULL): the fragment file -- its rows depend on nothing but their
 * reads either, so each batch's rows are formatted on the device and handed to the compressing threads as soon as the batch is traced;
 * kmahip_session_finish then only ends the file (its write_frag is ignored; with NULL here and write_frag set, the file is written at
 * the finish as <out_prefix>.frag.gz). ms[] of kmahip_session_finish: [1] the tracebacks, [3] nothing, [7] the fragment rows made
 * beside the batches. */
int kmahip_session_set_mt1(kmahip_session *s, int32_t tmpl, int one2one, const char *frag_path);
int kmahip_session_add(kmahip_session *s, const kmahip_read_batch *batch);
/* kmahip_session_add in two steps, for a caller whose reader thread is to go on while the device works: _upload returns when the
 * batch's host arrays are free again, _map runs stages 2 and 3a on what has been uploaded since the last call */
int kmahip_session_upload(kmahip_session *s, const kmahip_read_batch *batch);
int kmahip_session_map(kmahip_session *s);
int kmahip_session_finish(kmahip_session *s, const char *out_prefix, int write_fsa, int write_frag, int64_t *n_reads, int64_t *n_rows, double ms[8]);
void kmahip_session_close(kmahip_session *s);

/* Status of the *_dev calls issued on this workspace since the last query; synchronises `stream`. 0, or KMAHIP_EOVERFLOW
 * with kmahip_last_error() naming one of:
 *   - the library's own candidate pool ran out (stage 2): the pool has been doubled, repeat the scan call (and what
 *     followed it) -- no caller capacity is involved;
 *   - T_cap (kmahip_cands / kmahip_pe_recs) or ops_cap (kmahip_traces) too small: T_off[n] / R_off[2n] holds the needed
 *     size. A kmahip_align_*_dev call queued behind such a scan touches nothing beyond the capacities (it reports no hits),
 *     so scan_dev + align_dev may be chained on a stream without a status check in between;
 *   - more MEMs per (read, template) pair than the align scratch holds (a read full of repeats): the capacity has been raised
 *     fourfold, repeat the align (score vectors zeroed again) or trace call; the one-call runs (kmahip_run_*) do that themselves;
 *   - a DP problem beyond the trace scratch.
 * The status word is sticky until read here. */
int kmahip_ws_status(kmahip_ws *ws, void *stream);

/* Algorithmic work counters of the last scan on this workspace (read after a sync): probes = k-mer starts
 * whose value set was resolved (= hashMap_get calls of the reference), hash_probes = those that actually went
 * to the probe table (the rest were read off the template store while walking a match). */
typedef struct kmahip_scan_stats {
	uint64_t probes;
	uint64_t value_elems;
	uint64_t active_strands;
	uint64_t hash_probes;
	uint64_t prefilter_probes;   /* of probes / hash_probes: those issued by scan_prefilter_kernel */
} kmahip_scan_stats;
/* same for the last align call: template-index lookups, bases inside MEMs,
 * DP cells filled, (read, candidate) tasks aligned */
typedef struct kmahip_align_stats {
	uint64_t lookups;
	uint64_t mem_bases;
	uint64_t dp_cells;
	uint64_t tasks;
} kmahip_align_stats;
int kmahip_align_get_stats(kmahip_ws *ws, kmahip_align_stats *st, void *stream);
/* counting costs atomics in the kernel: off by default */
int kmahip_scan_set_stats(kmahip_ws *ws, int on);
int kmahip_scan_get_stats(kmahip_ws *ws, kmahip_scan_stats *st, void *stream);

/* work figures of the last long-read trace call on this workspace (kmahip_align_trace_mt1*, kmahip_run_mt1, or the trace stage
 * on reads over 1 kb): DP problems solved, their cells (rows x columns; rows x (band + 1) for banded ones), MEMs of the chained
 * strands, reads */
typedef struct kmahip_trace_stats {
	uint64_t problems, dp_cells, mems, reads;
} kmahip_trace_stats;
int kmahip_trace_get_stats(kmahip_ws *ws, kmahip_trace_stats *st);

/* Kernel timing: when on, every *_dev call records a HIP event pair around its
 * main kernels (scan_prefilter_kernel, scan_se_kernel, seed_tasks_kernel, align_tasks_kernel) on the caller's stream; get_timing waits
 * for them, returns the summed milliseconds and launch count, and resets. */
int kmahip_ws_set_timing(kmahip_ws *ws, int on);
int kmahip_ws_get_timing(kmahip_ws *ws, int kernel /* 0 scan_se_kernel, 1 align_tasks_kernel, 2 scan_prefilter_kernel, 3 seed_tasks_kernel */,
                         double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
