/* kma_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the KMA 1.5.1 seed-and-extend hot path, written
 * from the algorithm description in SURVEY.md / the reference sources (each
 * function cites the reference file:line whose behaviour it restates). It is
 * the checker that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg compare the HIP path against; nothing in kma_amd/ may link or call it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks this library
 * byte-for-byte against stream taps (-s2, frag_raw) produced by the compiled
 * reference (oracle/_ref/kma) and committed under tests/golden/.
 */
#ifndef KMA_ORACLE_H
#define KMA_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scoring constants, penalties.h:22-33, defaults kma.c:327-336,1307-1328 */
typedef struct {
	int M, MM, U, W1, Wl, Mn, PE;
	int d[5][5];
} orc_rewards;

/* in-memory image of <prefix>.comp.b (hashmapkma.h:26-48, App. A) plus
 * <prefix>.length.b and <prefix>.seq.b */
typedef struct {
	uint32_t DB_size, mlen, prefix_len, kmersize, flag;
	uint64_t prefix, size /* mask after load */, n, v_index, null_index;
	uint32_t *exist;      /* size+1 entries */
	uint32_t *key32;      /* n+1, when mlen <= 16 */
	uint64_t *key64;      /* n+1, when mlen > 16 */
	uint32_t *value_index;/* n */
	uint16_t *values16;   /* DB_size < 65535 */
	uint32_t *values32;
	/* template store */
	int32_t *tlen;        /* DB_size entries, tlen[0] = kmerindex */
	uint64_t *tseq;       /* concatenated 2-bit words */
	int64_t *tseq_off;    /* word offset of template i */
} orc_db;

void orc_default_rewards(orc_rewards *r);
orc_db *orc_db_load(const char *prefix);
void orc_db_free(orc_db *db);
/* returns value offset (element index into values) or -1 */
int64_t orc_hash_get(const orc_db *db, uint64_t key);

/* 2-bit codec, compdna.c:99-127, 228-256 */
void orc_pack(const uint8_t *codes, int len, uint64_t *seq, int *N /* N[0]=count */);
void orc_rc(const uint64_t *seq, int seqlen, const int *N, uint64_t *rseq, int *rN);

/* stage 2, -1t1 single end: savekmers.c:2442-3065.
 * seq must have complen+1 readable words. N = position list, nN entries.
 * Output: returns 1 if the read is emitted (mapped), 0 otherwise.
 * out_rc_flag = +-bestScore, out_flag = 0|16, T = template ids (negative =
 * reverse strand entries of a strand tie), *nT their count. emit_rc = 1 when
 * the reverse-complemented sequence is the one written to the S2 stream. */
int orc_scan_se(const orc_db *db, const orc_rewards *rw, int exhaustive,
                const uint64_t *seq, int seqlen, const int *N, int nN,
                int *out_rc_flag, int *out_flag, int *T, int *nT, int *emit_rc);

/* batched form over a CSR read set; T_off has n+1 entries on return.
 * Returns total number of template ids written (<= T_cap) or -needed. */
int64_t orc_scan_se_batch(const orc_db *db, const orc_rewards *rw, int exhaustive,
                          int64_t n_reads, const uint64_t *seq, const int64_t *seq_off,
                          const int32_t *len, const int32_t *N, const int64_t *N_off,
                          int32_t *rc_flag, int32_t *flag, int64_t *T_off,
                          int32_t *T, int64_t T_cap);

/* stage 2, paired end `-apm p`: save_kmers_penaltyPair (savekmers.c:3572-3777). out[0], out[1] are the
 * S2 records in stream order (present = 0: not written). mate = which input read the record carries,
 * rc = 1 when its reverse complement is the sequence written. T1/T2: caller buffers of 2*DB_size ints the
 * record lists point into. Returns the reference's unmapped mask (0 both written ... 3 none). */
/* stage 2, default mode (no -1t1): save_kmers_chain, savekmers.c:5127-5945. One read -> up to out_cap S2 records, one per
 * accepted chain: rc_flag = the chain's score (negative when both strands carry it: the reverse strand's templates follow as
 * negative ids), emit_rc = 1 when the reverse-complemented read is the one printed, [q_start, q_end) = the query bounds
 * appended to the header (qseqs.c:41-56). T lists live in T_pool. Returns the number of records, -1 when a capacity ran out. */
typedef struct { int rc_flag, emit_rc, nT, q_start, q_end; const int *T; } orc_chain_rec;
int orc_scan_chain(const orc_db *db, const orc_rewards *rw, int exhaustive, int minlen, double coverT, double mrs,
                   const uint64_t *seq, int seqlen, const int *N /* N[0] = count */, orc_chain_rec *out, int out_cap, int *T_pool, int T_cap);

typedef struct { int present, mate, rc, rc_flag, flag, nT; const int *T; } orc_pe_rec;
int orc_scan_pe(const orc_db *db, const orc_rewards *rw, int exhaustive,
                const uint64_t *seq1, int len1, const int *N1, int nN1,
                const uint64_t *seq2, int len2, const int *N2, int nN2,
                orc_pe_rec out[2], int *T1, int *T2);
/* the same for the union pairing (-apm u, and -ipe without -apm) */
int orc_scan_pe_union(const orc_db *db, const orc_rewards *rw, int exhaustive,
                const uint64_t *seq1, int len1, const int *N1, int nN1,
                const uint64_t *seq2, int len2, const int *N2, int nN2,
                orc_pe_rec out[2], int *T1, int *T2);
/* ... and for stage 2 of forced pairing (-apm f) */
int orc_scan_pe_force(const orc_db *db, const orc_rewards *rw, int exhaustive,
                const uint64_t *seq1, int len1, const int *N1, int nN1,
                const uint64_t *seq2, int len2, const int *N2, int nN2,
                orc_pe_rec out[2], int *T1, int *T2);

/* ---- stage 3a (oracle/align.c) ------------------------------------------ */
typedef struct orc_aligner orc_aligner;
typedef struct { int minlen, mq; double scoreT, mrc, minFrac; } orc_align_params;
orc_aligner *orc_aligner_new(const orc_db *db);
void orc_aligner_free(orc_aligner *a);
/* one read: alnFragsSE (alnfrags.c:1052-1218) + update_Scores (updatescores.c:203-298).
 * seq/N are the ORIGINAL read; flag & 16 selects its reverse complement as the
 * S2 stream would carry it. Outputs (capacity nT each): kept hits in candidate
 * order. n_hits = -1 marks a strand tie (rc_flag < 0), which is not restated. */
int orc_align_se(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap,
                 const uint64_t *seq, int seqlen, const int *N, int nN,
                 int rc_flag, int flag, const int *T, int nT,
                 int *n_hits, int *best_score, int *out_flag, int *ht, int *hs, int *he, int *hscore,
                 uint64_t *alignment_scores, uint64_t *uniq_alignment_scores);
int64_t orc_align_se_batch(const orc_db *db, const orc_rewards *rw, const orc_align_params *ap,
                           int64_t n_reads, const uint64_t *seq, const int64_t *seq_off,
                           const int32_t *len, const int32_t *N, const int64_t *N_off,
                           const int32_t *rc_flag, const int32_t *flag, const int64_t *T_off, const int32_t *T,
                           int32_t *n_hits, int32_t *best_score, int32_t *out_flag,
                           int32_t *ht, int32_t *hs, int32_t *he, int32_t *hscore,
                           uint64_t *alignment_scores, uint64_t *uniq_alignment_scores);
/* paired end stage 3a (alnFragsPenaltyPE). kind: 0 unmapped, 1 proper pair (hits shared by both mates),
 * 2 unmated pair, 3 first record only, 4 second record only. Arrays hold nT entries (caller-owned). */
typedef struct {
	int kind, swapped, n_hits, best, best_r, flagA, flagB, rcA, rcB;
	int n_hits_r;              /* kind 2: hits of the second record follow the first record's in the arrays */
	int *tmpl, *score, *start, *end;
} orc_pe_out;
int orc_align_pe(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap,
                 const uint64_t *seqA, int lenA, const int *NA, int nNA, int flagA,
                 const uint64_t *seqB, int lenB, const int *NB, int nNB, int flagB,
                 const int *T, int nT, orc_pe_out *out,
                 uint64_t *alignment_scores, uint64_t *uniq_alignment_scores);
/* stage 3c, per read: KMA() with traceback (align.c:214-507) + the read filter of assemble_KMA (assembly.c:1917-1965).
 * See align.c. */
int orc_align_trace(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap, const uint8_t *read, int q_len, int t,
                    int *stats, char *cols, int cap);
/* the `-Mt1 t` form of the same: raw read, strand picked by anker_rc (align.c:780-991). See align.c. */
int orc_align_trace_mt1(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap, uint8_t *read, int q_len, int t,
                        int one2one, int exhaustive, int *stats, char *cols, int cap, int *is_rc);
void orc_nw_tap(const uint64_t *tseq, int tlen_total, const uint8_t *q, int k, int t_s, int t_e, int q_s, int q_e,
                int band, const orc_rewards *rw, int out[6]);

/* stage 3b: ConClave template choice per frag_raw record (runConClave, conclave.c:43-215) and the leading
 * columns of the `.res` rows (runkma.c:608-613, 765-783; p_chisqr stdstat.c:136-147). See conclave.c. */
int orc_conclave(int64_t n_rec, const int32_t *n_hits, const int32_t *read_score, const int32_t *q_len, const int32_t *q_len2,
                 const int64_t *off, const int32_t *tmpl, const int32_t *start, const int32_t *end,
                 const uint64_t *alignment_scores, const uint64_t *uniq_alignment_scores, const int32_t *template_lengths,
                 int32_t *out_tmpl, int32_t *out_start, int32_t *out_end,
                 uint64_t *w_scores, uint32_t *fragmentCounts, uint32_t *readCounts, uint64_t *depth);
double orc_p_chisqr(long double q);
int orc_res_stats(int DB_size, const uint64_t *w_scores, const int32_t *template_lengths, double evalue, double scoreT,
                  double *expected, double *q_value, double *p_value, int32_t *significant);

/* stage 3c per template: pile-up (alnToMat, assembly.c:1317-1444), consensus (callConsensus :1499-1631) -- see assembly.c */
typedef struct orc_assembly orc_assembly;
orc_assembly *orc_assembly_new(int t_len);
void orc_assembly_free(orc_assembly *m);
void orc_assembly_add(orc_assembly *m, const char *cols, int aln_len, const uint8_t *read, int start, int score);
void orc_assembly_call(const orc_assembly *m, const uint64_t *tseq, int bcd, double evalue, int64_t *out, char *cons);
/* caller: 0 baseCaller (default), 1 nanoCaller (-bcNano, assembly.c:205-240); sig: 0 significantNuc, 1 significantAnd90Nuc
 * (-bcNano, assembly.c:147-149, kma.c:762-766) */
void orc_assembly_call2(const orc_assembly *m, const uint64_t *tseq, int bcd, double evalue, int caller, int sig, int64_t *out, char *cons);
int orc_assembly_has_reads(const orc_assembly *m);
const uint64_t *orc_db_template(const orc_db *db, int t);

#ifdef __cplusplus
}
#endif
#endif
