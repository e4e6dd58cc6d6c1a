/* oracle/scan.c -- TEST INFRASTRUCTURE (see kma_oracle.h).
 * CPU restatement of stage 2 for `-1t1` single-end reads:
 *   2-bit codec          compdna.c:99-127 (pack), :214-256 (binRev, rc_comp)
 *   k-mer extraction     stdnuc.h:20-30
 *   save_kmers           savekmers.c:2442-3065
 *   getBestMatch         savekmers.c:273-294
 * Written as a per-position event stream (hit with value-set id / miss), which
 * is the formulation the HIP kernel uses; N handling is folded into "every
 * k-mer window touching an N is a miss" (equivalent to savekmers.c:2704).
 */
#include "kma_oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_pack(const uint8_t *codes, int len, uint64_t *seq, int *N) {
	/* compdna.c:99-127: 32 bases per word, first base in the top bits, N
	 * stored as 0 with its position appended to N[1..N[0]], last word
	 * left-aligned. */
	int words = (len + 31) >> 5;
	N[0] = 0;
	for(int w = 0; w < words; ++w) {
		uint64_t x = 0;
		int end = (w << 5) + 32 < len ? (w << 5) + 32 : len;
		for(int j = w << 5; j < end; ++j) {
			x <<= 2;
			if(codes[j] == 4) N[++N[0]] = j; else x |= codes[j] & 3;
		}
		if(end & 31) x <<= 64 - ((end & 31) << 1);
		seq[w] = x;
	}
}

static uint64_t rev2(uint64_t x) {
	/* reverse the order of the 32 two-bit symbols (compdna.c:214-226) */
	x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
	x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
	x = ((x >> 8) & 0x00FF00FF00FF00FFull) | ((x & 0x00FF00FF00FF00FFull) << 8);
	x = ((x >> 16) & 0x0000FFFF0000FFFFull) | ((x & 0x0000FFFF0000FFFFull) << 16);
	return (x >> 32) | (x << 32);
}

void orc_rc(const uint64_t *seq, int seqlen, const int *N, uint64_t *rseq, int *rN) {
	/* compdna.c:228-256 */
	int words = (seqlen + 31) >> 5;
	for(int i = 0; i < words; ++i) rseq[words - 1 - i] = rev2(~seq[i]);
	if(seqlen & 31) {
		int sh = ((words << 5) - seqlen) << 1;
		for(int i = 0; i + 1 < words; ++i) rseq[i] = (rseq[i] << sh) | (rseq[i + 1] >> (64 - sh));
		rseq[words - 1] <<= sh;
	}
	rN[0] = N[0];
	for(int i = 1; i <= N[0]; ++i) rN[i] = seqlen - 1 - N[N[0] + 1 - i];
}

static inline uint64_t kmer_at(const uint64_t *seq, int pos, int k) {
	/* stdnuc.h:27-30 (getKmer_macro); may touch seq[word+1] */
	int ip = (pos & 31) << 1, w = pos >> 5, sh = 64 - (k << 1);
	if(ip <= sh) return (seq[w] << ip) >> sh;
	return ((seq[w] << ip) | (seq[w + 1] >> (64 - ip))) >> sh;
}

/* score for bridging `gaps` missed k-mer starts between two hits of one
 * template (savekmers.c:2527-2569 run form / :2590-2627 per-template form).
 * first_of_run: cost of the hit k-mer itself is included (M per base of the
 * bridged window) exactly as the reference splits it. */
typedef struct { int Ms, MMs, Us, W1s; } runacc;

static void bridge_same_set(runacc *a, int gaps, int k, int mlen, const orc_rewards *rw) {
	/* savekmers.c:2522-2569 */
	if(gaps == 0) {
		a->Ms += 1;
	} else if(mlen <= gaps && gaps <= k) {
		a->Ms += k; a->MMs += 1;
	} else if(k < gaps) {
		int mm, m;
		a->Ms += k;
		gaps -= (k - 1);
		if(gaps <= 2) { mm = gaps; m = 0; }
		else {
			mm = gaps / k + (gaps % k ? 1 : 0);
			if(mm < 2) mm = 2;
			m = gaps - mm; if(k < m) m = k; if(mm < m) m = mm;
		}
		if(rw->W1 + (gaps - 1) * rw->U <= mm * rw->MM + m * rw->M) { a->MMs += mm; a->Ms += m; }
		else { a->W1s += 1; a->Us += gaps - 1; }
	} else if(mlen != k) {
		a->Ms += gaps; a->MMs += 1;
	} else {
		a->Ms += gaps; a->W1s += 1; a->Us += k - gaps;
	}
}

static int bridge_new_set(int gaps, int k, int mlen, int is_rc, const orc_rewards *rw) {
	/* savekmers.c:2590-2627 (forward) / :2901-2938 (reverse; SNP branch
	 * differs: k*M+MM instead of gaps*M+MM) */
	if(gaps == 0) return rw->M;
	if(mlen <= gaps && gaps <= k) return (is_rc ? k : gaps) * rw->M + rw->MM;
	if(k < gaps) {
		int mm, m;
		gaps -= (k - 1);
		if(gaps <= 2) { mm = gaps; m = 0; }
		else {
			mm = gaps / k + (gaps % k ? 1 : 0);
			if(mm < 2) mm = 2;
			m = gaps - mm; if(k < m) m = k; if(mm < m) m = mm;
		}
		int sub = mm * rw->MM + m * rw->M, ind = rw->W1 + (gaps - 1) * rw->U;
		return k * rw->M + (ind <= sub ? sub : ind);
	}
	if(mlen != k) return gaps * rw->M + rw->MM;
	return gaps * rw->M + (k - gaps) * rw->U + rw->W1;
}

typedef struct {
	int *score;      /* DB_size */
	int *ext;        /* DB_size: position of the template's last hit */
	uint8_t *incl;   /* DB_size */
} dense_state;

static inline int vcount(const orc_db *db, int64_t v) { return db->values16 ? db->values16[v] : (int) db->values32[v]; }
static inline int vat(const orc_db *db, int64_t v, int i) { return db->values16 ? db->values16[v + i] : (int) db->values32[v + i]; }

/* one strand: returns best score, best[] = tied templates in first-seen order */
static int scan_strand(const orc_db *db, const orc_rewards *rw, int exhaustive, int is_rc,
                       const uint64_t *seq, int seqlen, const int *N /* N[0]=count, list, no sentinel */,
                       dense_state *st, int *list, int *nbest) {
	const int k = db->kmersize, mlen = db->mlen;
	const int npos = seqlen - k + 1;
	int nN = N[0];
	*nbest = 0;

	/* prefilter, savekmers.c:2477-2495: every k-th k-mer of each N-free
	 * segment, stop at first hit */
	int hit = exhaustive;
	{
		int j = 0;
		for(int i = 1; i <= nN + 1 && !hit; ++i) {
			int segend = (i <= nN) ? N[i] : seqlen;
			for(; j < segend - k + 1 && !hit; j += k) {
				if(orc_hash_get(db, kmer_at(seq, j, k)) >= 0) hit = 1;
			}
			j = segend + 1;
		}
	}
	if(!hit) return 0;

	int nlist = 0, hitCounter = 0, gaps = 0, HIT = 0;
	int64_t last = -1;
	runacc acc = {0, 0, 0, 0};
	int ni = 1;                    /* next N index */
	int blocked_until = -1;        /* positions <= this overlap an N */
	for(int p = 0; p < npos; ++p) {
		/* window [p, p+k) touches an N ? */
		while(ni <= nN && N[ni] < p) ++ni;
		int miss = (ni <= nN && N[ni] < p + k);
		(void) blocked_until;
		int64_t v = miss ? -1 : orc_hash_get(db, kmer_at(seq, p, k));
		if(v < 0) { ++gaps; continue; }
		if(v == last) {
			bridge_same_set(&acc, gaps, k, mlen, rw);
			HIT = p; gaps = 0;
		} else {
			if(last >= 0) {
				int sc = acc.Ms * rw->M + acc.MMs * rw->MM + acc.Us * rw->U + acc.W1s * rw->W1;
				int c = vcount(db, last);
				for(int i = 1; i <= c; ++i) { int t = vat(db, last, i); st->score[t] += sc; st->ext[t] = HIT; }
				HIT = p - 1;
				last = v;
				c = vcount(db, v);
				for(int i = 1; i <= c; ++i) {
					int t = vat(db, v, i);
					if(st->incl[t]) {
						st->score[t] += bridge_new_set(HIT - st->ext[t], k, mlen, is_rc, rw);
					} else {
						st->score[t] = k * rw->M;
						st->incl[t] = 1;
						list[nlist++] = t;
					}
				}
			} else {
				last = v;
				int c = vcount(db, v);
				for(int i = 1; i <= c; ++i) {
					int t = vat(db, v, i);
					st->score[t] = k * rw->M; st->incl[t] = 1; list[i - 1] = t;
				}
				nlist = c;
			}
			HIT = p; gaps = 0;
			acc.Ms = acc.MMs = acc.Us = acc.W1s = 0;
		}
		++hitCounter;
	}
	if(last >= 0) {
		int sc = acc.Ms * rw->M + acc.MMs * rw->MM + acc.Us * rw->U + acc.W1s * rw->W1;
		int c = vcount(db, last);
		for(int i = 1; i <= c; ++i) st->score[vat(db, last, i)] += sc;
	}
	/* clean-up + clamp (savekmers.c:2744-2753) and getBestMatch (:273-294) */
	int best = 0, nb = 0;
	for(int i = 0; i < nlist; ++i) {
		int t = list[i];
		int s = st->score[t] < 0 ? 0 : st->score[t];
		st->score[t] = 0; st->ext[t] = 0; st->incl[t] = 0;
		if(!hitCounter) continue;
		if(s > best) { best = s; nb = 0; list[nb++] = t; }
		else if(s == best) list[nb++] = t;
	}
	*nbest = nb;
	return best;
}

typedef struct {
	dense_state st;
	int *fwd, *rev;
	uint64_t *rseq; int rcap;
	int *rN; int ncap;
	uint32_t DB_size;
} scan_ws;

static scan_ws *ws_new(const orc_db *db) {
	scan_ws *w = calloc(1, sizeof *w);
	w->DB_size = db->DB_size;
	w->st.score = calloc(db->DB_size + 1, sizeof(int));
	w->st.ext = calloc(db->DB_size + 1, sizeof(int));
	w->st.incl = calloc(db->DB_size + 1, 1);
	w->fwd = malloc((2 * (size_t) db->DB_size + 4) * sizeof(int));
	w->rev = malloc(((size_t) db->DB_size + 4) * sizeof(int));
	return w;
}
static void ws_free(scan_ws *w) {
	free(w->st.score); free(w->st.ext); free(w->st.incl); free(w->fwd); free(w->rev);
	free(w->rseq); free(w->rN); free(w);
}

static int scan_se_ws(scan_ws *w, const orc_db *db, const orc_rewards *rw, int exhaustive,
                      const uint64_t *seq, int seqlen, const int *N, int nN,
                      int *out_rc_flag, int *out_flag, int *T, int *nT, int *emit_rc) {
	const int k = db->kmersize;
	*nT = 0; *out_rc_flag = 0; *out_flag = 0; *emit_rc = 0;
	if(seqlen < k) return 0;              /* savekmers.c:2452 */
	int words = (seqlen + 31) >> 5;
	if(w->rcap < words + 2) { free(w->rseq); w->rcap = words + 66; w->rseq = calloc(w->rcap, 8); }
	if(w->ncap < nN + 2) { free(w->rN); w->ncap = nN + 66; w->rN = malloc(2 * (size_t) w->ncap * sizeof(int)); }
	int *fN = w->rN + w->ncap;            /* forward list with count in [0] */
	fN[0] = nN; memcpy(fN + 1, N, (size_t) nN * sizeof(int));
	orc_rc(seq, seqlen, fN, w->rseq, w->rN);
	w->rseq[words] = 0;

	int nf = 0, nr = 0;
	int bs = scan_strand(db, rw, exhaustive, 0, seq, seqlen, fN, &w->st, w->fwd, &nf);
	int br = scan_strand(db, rw, exhaustive, 1, w->rseq, seqlen, w->rN, &w->st, w->rev, &nr);
	/* savekmers.c:3037-3062 */
	if(!(bs > 0 || br > 0) || !(k <= bs || k <= br)) return 0;
	if(bs > br) {
		memcpy(T, w->fwd, (size_t) nf * sizeof(int)); *nT = nf; *out_rc_flag = bs; *out_flag = 0;
	} else if(bs < br) {
		memcpy(T, w->rev, (size_t) nr * sizeof(int)); *nT = nr; *out_rc_flag = br; *out_flag = 16; *emit_rc = 1;
	} else {
		memcpy(T, w->fwd, (size_t) nf * sizeof(int));
		for(int i = 0; i < nr; ++i) T[nf + i] = -w->rev[i];
		*nT = nf + nr; *out_rc_flag = -bs; *out_flag = 0;
	}
	return 1;
}

int orc_scan_se(const orc_db *db, const orc_rewards *rw, int exhaustive,
                const uint64_t *seq, int seqlen, const int *N, int nN,
                int *out_rc_flag, int *out_flag, int *T, int *nT, int *emit_rc) {
	scan_ws *w = ws_new(db);
	int r = scan_se_ws(w, db, rw, exhaustive, seq, seqlen, N, nN, out_rc_flag, out_flag, T, nT, emit_rc);
	ws_free(w);
	return r;
}

int64_t orc_scan_se_batch(const orc_db *db, const orc_rewards *rw, int exhaustive,
                          int64_t n_reads, const uint64_t *seq, const int64_t *seq_off,
                          const int32_t *len, const int32_t *N, const int64_t *N_off,
                          int32_t *rc_flag, int32_t *flag, int64_t *T_off,
                          int32_t *T, int64_t T_cap) {
	scan_ws *w = ws_new(db);
	int *tmp = malloc((2 * (size_t) db->DB_size + 4) * sizeof(int));
	int64_t total = 0;
	int overflow = 0;
	T_off[0] = 0;
	for(int64_t r = 0; r < n_reads; ++r) {
		int nT = 0, rf = 0, fl = 0, erc = 0;
		int words = (len[r] + 31) >> 5;
		/* private copy with a pad word (getKmer may read word+1) */
		uint64_t *s = malloc(((size_t) words + 2) * 8);
		memcpy(s, seq + seq_off[r], (size_t) words * 8); s[words] = 0; s[words + 1] = 0;
		int mapped = scan_se_ws(w, db, rw, exhaustive, s, len[r], N + N_off[r], (int) (N_off[r + 1] - N_off[r]),
		                        &rf, &fl, tmp, &nT, &erc);
		free(s);
		if(!mapped) nT = 0;
		rc_flag[r] = mapped ? rf : 0;
		flag[r] = mapped ? fl : 0;
		if(total + nT <= T_cap) memcpy(T + total, tmp, (size_t) nT * sizeof(int)); else overflow = 1;
		total += nT;
		T_off[r + 1] = total;
	}
	free(tmp);
	ws_free(w);
	return overflow ? -total : total;
}
