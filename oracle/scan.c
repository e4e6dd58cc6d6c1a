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
static int scan_strand_x(const orc_db *db, const orc_rewards *rw, int exhaustive, int is_rc,
                         const uint64_t *seq, int seqlen, const int *N /* N[0]=count, list, no sentinel */,
                         dense_state *st, int *list, int *nbest, int *full_scores, int *hit_count) {
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
	if(!hit) { if(hit_count) *hit_count = 0; return 0; }

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
	if(hit_count) *hit_count = hitCounter;
	if(full_scores) {
		/* get_kmers_for_pair (savekmers.c:655-677): every candidate with its clamped score, first-seen order */
		for(int i = 0; i < nlist; ++i) {
			int t = list[i];
			full_scores[i] = st->score[t] < 0 ? 0 : st->score[t];
			st->score[t] = 0; st->ext[t] = 0; st->incl[t] = 0;
		}
		*nbest = nlist;
		return hitCounter;
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

static int scan_strand(const orc_db *db, const orc_rewards *rw, int exhaustive, int is_rc,
                       const uint64_t *seq, int seqlen, const int *N, dense_state *st, int *list, int *nbest) {
	return scan_strand_x(db, rw, exhaustive, is_rc, seq, seqlen, N, st, list, nbest, 0, 0);
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

/* ---- paired end, `-apm p`: save_kmers_penaltyPair (savekmers.c:3572-3777) -------------------
 * per mate: get_kmers_for_pair (:427-688) = both strands scored like save_kmers, ALL candidates kept;
 * getFirstPen (:1383), getSecondBestPen (:1415), getF_Best (:1648). `rev` = 1 (no prefix index). */
typedef struct { int n; int *t; int *s; } plist;

static int plist_find(const plist *l, int t) { for(int i = 0; i < l->n; ++i) if(l->t[i] == t) return l->s[i]; return 0; }

int orc_scan_pe(const orc_db *db, const orc_rewards *rw, int exhaustive,
                const uint64_t *seq1, int len1, const int *N1, int nN1,
                const uint64_t *seq2, int len2, const int *N2, int nN2,
                orc_pe_rec out[2], int *T1, int *T2) {
	const int k = db->kmersize, D = db->DB_size;
	scan_ws *w = ws_new(db);
	int *buf = malloc(sizeof(int) * (size_t) (8 * D + 32));
	plist F1 = {0, buf, buf + D}, R1 = {0, buf + 2 * D, buf + 3 * D}, F2 = {0, buf + 4 * D, buf + 5 * D}, R2 = {0, buf + 6 * D, buf + 7 * D};
	int hc1 = 0, hc2 = 0;
	for(int m = 0; m < 2; ++m) {
		const uint64_t *seq = m ? seq2 : seq1; const int len = m ? len2 : len1; const int *N = m ? N2 : N1; const int nN = m ? nN2 : nN1;
		plist *F = m ? &F2 : &F1, *R = m ? &R2 : &R1;
		if(len < k) continue;
		const int words = (len + 31) >> 5;
		uint64_t *rs = calloc((size_t) words + 2, 8);
		int *fN = malloc(sizeof(int) * (size_t) (2 * nN + 4)), *rN = fN + nN + 2;
		fN[0] = nN; memcpy(fN + 1, N, sizeof(int) * (size_t) nN);
		orc_rc(seq, len, fN, rs, rN);
		int hf = 0, hr = 0;
		scan_strand_x(db, rw, exhaustive, 0, seq, len, fN, &w->st, F->t, &F->n, F->s, &hf);
		scan_strand_x(db, rw, exhaustive, 1, rs, len, rN, &w->st, R->t, &R->n, R->s, &hr);
		if(m) hc2 = hf > hr ? hf : hr; else hc1 = hf > hr ? hf : hr;
		free(rs); free(fN);
	}
	/* mate 1: getFirstPen -> region = [F1 (+), R1 (-)] with scores */
	int nreg = 0, *regT = T1, *regS = malloc(sizeof(int) * (size_t) (2 * D + 4));
	int best1 = 0;
	if(hc1) {
		for(int i = 0; i < F1.n; ++i) { if(best1 < F1.s[i]) best1 = F1.s[i]; regT[nreg] = F1.t[i]; regS[nreg++] = F1.s[i]; }
		for(int i = 0; i < R1.n; ++i) { if(best1 < R1.s[i]) best1 = R1.s[i]; regT[nreg] = -R1.t[i]; regS[nreg++] = R1.s[i]; }
	}
	int paired = 0, best2 = 0, nb2 = 0, *bT = T2;
	if(hc2) {
		if(0 < best1) {
			/* getSecondBestPen */
			for(int i = 0; i < F2.n; ++i) { if(best2 < F2.s[i]) best2 = F2.s[i]; bT[nb2++] = F2.t[i]; }
			for(int i = 0; i < R2.n; ++i) { if(best2 < R2.s[i]) best2 = R2.s[i]; bT[nb2++] = -R2.t[i]; }
			int hits = 0;
			if(best2) {
				int comp = best1 + best2 - rw->PE; if(comp < 0) comp = 0;
				for(int i = 0; i < nreg; ++i) {
					int sc = regT[i] > 0 ? plist_find(&R2, regT[i]) : plist_find(&F2, -regT[i]);
					if(0 < sc) {
						sc += regS[i];
						if(comp < sc) { comp = sc; hits = 1; regT[0] = regT[i]; }
						else if(comp == sc) { regT[hits++] = regT[i]; }
					}
				}
			}
			if(hits) { paired = 1; nreg = hits; }
			else {
				int c = 0;
				for(int i = 0; i < nreg; ++i) if(best1 == regS[i]) regT[c++] = regT[i];
				nreg = c; c = 0;
				for(int i = 0; i < nb2; ++i) {
					int t = bT[i];
					if(0 < t) { if(best2 == plist_find(&F2, t)) bT[c++] = t; }
					else { if(best2 <= plist_find(&R2, -t)) bT[c++] = t; }
				}
				nb2 = c;
			}
		} else {
			/* getF_Best on mate 2 -> region */
			nreg = 0;
			for(int i = 0; i < F2.n; ++i) { int sc = F2.s[i]; if(best2 < sc) { best2 = sc; nreg = 0; regT[nreg++] = F2.t[i]; } else if(best2 == sc) regT[nreg++] = F2.t[i]; }
			for(int i = 0; i < R2.n; ++i) { int sc = R2.s[i]; if(best2 < sc) { best2 = sc; nreg = 0; regT[nreg++] = -R2.t[i]; } else if(best2 == sc) regT[nreg++] = -R2.t[i]; }
		}
	}
	/* both mates are left reverse-complemented by get_kmers_for_pair when they were scanned */
	int o1 = len1 >= k, o2 = len2 >= k;
	int flag = 65, flag_r = 129, ret = 3;
	memset(out, 0, 2 * sizeof(orc_pe_rec));
	#define EMIT(slot, MATE, RC, SCORE, FLAG, TP, NT) do { out[slot].present = 1; out[slot].mate = MATE; out[slot].rc = RC; \
		out[slot].rc_flag = SCORE; out[slot].flag = FLAG; out[slot].nT = NT; out[slot].T = TP; } while(0)
	if(0 < best1 && 0 < best2) {
		if(paired) {
			flag |= 2; flag_r |= 2;
			int comp = (hc1 + hc2) < (best1 + best2) ? (hc1 + hc2) : (best1 + best2);
			if(k <= comp || (unsigned) (len1 + len2 - comp - (k << 1)) < (unsigned) (comp * k)) {   /* CompDNA.seqlen is unsigned: the difference wraps */
				if(0 < regT[0]) {
					flag |= 32; flag_r |= 16; o1 ^= 1;
					EMIT(0, 0, o1, best1, flag, regT, 0);
					EMIT(1, 1, o2, best2, flag_r, regT, nreg);
				} else {
					flag |= 16; flag_r |= 32; o2 ^= 1;
					for(int i = 0; i < nreg; ++i) regT[i] = -regT[i];
					EMIT(0, 1, o2, best2, flag_r, regT, 0);
					EMIT(1, 0, o1, best1, flag, regT, nreg);
				}
				ret = 0;
			}
		} else {
			int h1 = hc1 < best1 ? hc1 : best1, h2 = hc2 < best2 ? hc2 : best2;
			int ok1 = k <= h1 || (unsigned) (len1 - h1 - k) < (unsigned) (h1 * k), ok2 = k <= h2 || (unsigned) (len2 - h2 - k) < (unsigned) (h2 * k);
			int s1 = best1, s2 = best2;
			if(ok1) {
				if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
				else { flag |= 16; flag_r |= 32; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			}
			if(ok2) {
				if(0 < bT[0]) { o2 ^= 1; if(bT[nb2 - 1] < 0) s2 = -s2; }
				else { flag |= 32; flag_r |= 16; for(int i = 0; i < nb2; ++i) bT[i] = -bT[i]; }
			}
			if(ok1) EMIT(0, 0, o1, s1, flag, regT, nreg);
			if(ok2) EMIT(1, 1, o2, s2, flag_r, bT, nb2);
			ret = (ok1 ? 0 : 1) + (ok2 ? 0 : 2);
		}
	} else if(0 < best1) {
		int h1 = hc1 < best1 ? hc1 : best1, s1 = best1;
		if(k <= h1 || (unsigned) (len1 - h1 - k) < (unsigned) (h1 * k)) {
			flag |= 8; flag |= 32;
			if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
			else { flag |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			EMIT(0, 0, o1, s1, flag, regT, nreg);
			ret = 2;
		}
	} else if(0 < best2) {
		int h2 = hc2 < best2 ? hc2 : best2, s2 = best2;
		if(k <= h2 || (unsigned) (len2 - h2 - k) < (unsigned) (h2 * k)) {
			flag_r |= 8; flag_r |= 32;
			if(0 < regT[0]) { o2 ^= 1; if(regT[nreg - 1] < 0) s2 = -s2; }
			else { flag_r |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			EMIT(1, 1, o2, s2, flag_r, regT, nreg);
			ret = 1;
		}
	}
	#undef EMIT
	free(regS); free(buf);
	ws_free(w);
	return ret;
}

/* `-apm f`: save_kmers_forcePair (savekmers.c:3779-3864) with getFirstForce (:1254) and getSecondBestForce (:1275). Mate 1's candidates
 * of both strands go into the region list with their scores; mate 2 is scanned only if mate 1 had a hit; a region template that mate 2
 * hits on the OTHER strand scores the sum, the best sums stay (in the order met). A couple is printed when the best sum reaches k or
 * covers enough of the two reads -- both records with the sum, negated when the list ends on a reverse template (decided BEFORE the
 * list is turned for a couple that starts on a reverse template) -- and nothing is printed otherwise. Same in / out as orc_scan_pe. */
int orc_scan_pe_force(const orc_db *db, const orc_rewards *rw, int exhaustive,
                      const uint64_t *seq1, int len1, const int *N1, int nN1,
                      const uint64_t *seq2, int len2, const int *N2, int nN2,
                      orc_pe_rec out[2], int *T1, int *T2) {
	const int k = db->kmersize, D = db->DB_size;
	(void) T2;
	scan_ws *w = ws_new(db);
	int *buf = malloc(sizeof(int) * (size_t) (8 * D + 32));
	plist F1 = {0, buf, buf + D}, R1 = {0, buf + 2 * D, buf + 3 * D}, F2 = {0, buf + 4 * D, buf + 5 * D}, R2 = {0, buf + 6 * D, buf + 7 * D};
	int hc1 = 0, hc2 = 0;
	for(int m = 0; m < 2; ++m) {
		const uint64_t *seq = m ? seq2 : seq1; const int len = m ? len2 : len1; const int *N = m ? N2 : N1; const int nN = m ? nN2 : nN1;
		plist *F = m ? &F2 : &F1, *R = m ? &R2 : &R1;
		if(len < k || (m && !hc1)) continue;          /* (savekmers.c:3797: without a hit of mate 1 the function returns before mate 2 is looked at) */
		const int words = (len + 31) >> 5;
		uint64_t *rs = calloc((size_t) words + 2, 8);
		int *fN = malloc(sizeof(int) * (size_t) (2 * nN + 4)), *rN = fN + nN + 2;
		fN[0] = nN; memcpy(fN + 1, N, sizeof(int) * (size_t) nN);
		orc_rc(seq, len, fN, rs, rN);
		int hf = 0, hr = 0;
		scan_strand_x(db, rw, exhaustive, 0, seq, len, fN, &w->st, F->t, &F->n, F->s, &hf);
		scan_strand_x(db, rw, exhaustive, 1, rs, len, rN, &w->st, R->t, &R->n, R->s, &hr);
		if(m) hc2 = hf > hr ? hf : hr; else hc1 = hf > hr ? hf : hr;
		free(rs); free(fN);
	}
	memset(out, 0, 2 * sizeof(orc_pe_rec));
	int ret = 3;
	if(hc1 && hc2) {
		/* getFirstForce: the region list; getSecondBestForce over it */
		int nreg = 0, *regT = T1, *regS = malloc(sizeof(int) * (size_t) (2 * D + 4));
		for(int i = 0; i < F1.n; ++i) { regT[nreg] = F1.t[i]; regS[nreg++] = F1.s[i]; }
		for(int i = 0; i < R1.n; ++i) { regT[nreg] = -R1.t[i]; regS[nreg++] = R1.s[i]; }
		int best = 0, hits = 0;
		for(int i = 0; i < nreg; ++i) {
			const int rt = regT[i];
			int sc = rt > 0 ? plist_find(&R2, rt) : plist_find(&F2, -rt);
			if(!sc) continue;
			sc += regS[i];
			if(best < sc) { best = sc; hits = 1; regT[0] = rt; }
			else if(best == sc) regT[hits++] = rt;
		}
		if(best && (k <= best || (unsigned) len1 + (unsigned) len2 - (unsigned) best < (unsigned) (best * k))) {
			int o1 = len1 >= k, o2 = len2 >= k, flag = 67, flag_r = 131;
			const int s = regT[hits - 1] < 0 ? -best : best;
			#define EMIT(slot, MATE, RC, SCORE, FLAG, TP, NT) do { out[slot].present = 1; out[slot].mate = MATE; out[slot].rc = RC; \
				out[slot].rc_flag = SCORE; out[slot].flag = FLAG; out[slot].nT = NT; out[slot].T = TP; } while(0)
			if(0 < regT[0]) {
				flag |= 32; flag_r |= 16; o1 ^= 1;
				EMIT(0, 0, o1, s, flag, regT, 0);
				EMIT(1, 1, o2, s, flag_r, regT, hits);
			} else {
				flag |= 16; flag_r |= 32; o2 ^= 1;
				for(int i = 0; i < hits; ++i) regT[i] = -regT[i];
				EMIT(0, 1, o2, s, flag_r, regT, 0);
				EMIT(1, 0, o1, s, flag, regT, hits);
			}
			#undef EMIT
			ret = 0;
		}
		free(regS);
	}
	free(buf);
	ws_free(w);
	return ret;
}

/* `-apm u`, and what `-ipe` means without -apm (kma.c:206): save_kmers_unionPair (savekmers.c:3367-3570) with getF_Best / getR_Best
 * (:1648-1762). Mate 1 keeps the templates of either strand that reach its best score; mate 2 its own best set -- and where a template
 * of mate 1's set is in mate 2's set on the OTHER strand (getR_Best leaves a score standing only for mate 2's best ones, so "is
 * there a score" = "has the best score") those move to the front of mate 1's list and the two are printed as a couple; otherwise
 * each mate is a record of its own. Same in / out as orc_scan_pe. */
int orc_scan_pe_union(const orc_db *db, const orc_rewards *rw, int exhaustive,
                      const uint64_t *seq1, int len1, const int *N1, int nN1,
                      const uint64_t *seq2, int len2, const int *N2, int nN2,
                      orc_pe_rec out[2], int *T1, int *T2) {
	const int k = db->kmersize, D = db->DB_size;
	scan_ws *w = ws_new(db);
	int *buf = malloc(sizeof(int) * (size_t) (8 * D + 32));
	plist F1 = {0, buf, buf + D}, R1 = {0, buf + 2 * D, buf + 3 * D}, F2 = {0, buf + 4 * D, buf + 5 * D}, R2 = {0, buf + 6 * D, buf + 7 * D};
	int hc1 = 0, hc2 = 0;
	for(int m = 0; m < 2; ++m) {
		const uint64_t *seq = m ? seq2 : seq1; const int len = m ? len2 : len1; const int *N = m ? N2 : N1; const int nN = m ? nN2 : nN1;
		plist *F = m ? &F2 : &F1, *R = m ? &R2 : &R1;
		if(len < k) continue;
		const int words = (len + 31) >> 5;
		uint64_t *rs = calloc((size_t) words + 2, 8);
		int *fN = malloc(sizeof(int) * (size_t) (2 * nN + 4)), *rN = fN + nN + 2;
		fN[0] = nN; memcpy(fN + 1, N, sizeof(int) * (size_t) nN);
		orc_rc(seq, len, fN, rs, rN);
		int hf = 0, hr = 0;
		scan_strand_x(db, rw, exhaustive, 0, seq, len, fN, &w->st, F->t, &F->n, F->s, &hf);
		scan_strand_x(db, rw, exhaustive, 1, rs, len, rN, &w->st, R->t, &R->n, R->s, &hr);
		if(m) hc2 = hf > hr ? hf : hr; else hc1 = hf > hr ? hf : hr;
		free(rs); free(fN);
	}
	/* getF_Best: the best score over both strands' candidates, its templates in the order met, the forward strand's first */
	#define BEST_OF(F, R, DST, CNT, BEST) do { BEST = 0; CNT = 0; \
		for(int i_ = 0; i_ < (F).n; ++i_) { const int sc_ = (F).s[i_]; if(BEST < sc_) { BEST = sc_; CNT = 0; DST[CNT++] = (F).t[i_]; } else if(BEST == sc_) DST[CNT++] = (F).t[i_]; } \
		for(int i_ = 0; i_ < (R).n; ++i_) { const int sc_ = (R).s[i_]; if(BEST < sc_) { BEST = sc_; CNT = 0; DST[CNT++] = -(R).t[i_]; } else if(BEST == sc_) DST[CNT++] = -(R).t[i_]; } } while(0)
	int *regT = T1, *bT = T2, nreg = 0, nb2 = 0, best1 = 0, best2 = 0, paired = 0;
	if(hc1) {
		BEST_OF(F1, R1, regT, nreg, best1);
		if(k < best1 && (unsigned) (best1 * k) < (unsigned) len1 - (unsigned) best1) best1 = 0;      /* (CompDNA.seqlen is unsigned) */
	}
	if(hc2) {
		if(best1) {
			BEST_OF(F2, R2, bT, nb2, best2);
			int hits = 0;
			if(0 < best2) {
				for(int i = 0; i < nreg; ++i) {
					const int rt = regT[i];
					const int sc = rt > 0 ? plist_find(&R2, rt) : plist_find(&F2, -rt);
					if(sc == best2) { const int x = regT[hits]; regT[hits] = rt; regT[i] = x; ++hits; }
				}
			}
			if(hits) { paired = 1; nreg = hits; }
		} else BEST_OF(F2, R2, regT, nreg, best2);
		if(k < best2 && (unsigned) (best2 * k) < (unsigned) len2 - (unsigned) best2) { best2 = 0; paired = 0; }
	}
	#undef BEST_OF
	int o1 = len1 >= k, o2 = len2 >= k;          /* get_kmers_for_pair leaves a scanned mate reverse-complemented */
	int flag = 65, flag_r = 129, ret = 3;
	memset(out, 0, 2 * sizeof(orc_pe_rec));
	#define EMIT(slot, MATE, RC, SCORE, FLAG, TP, NT) do { out[slot].present = 1; out[slot].mate = MATE; out[slot].rc = RC; \
		out[slot].rc_flag = SCORE; out[slot].flag = FLAG; out[slot].nT = NT; out[slot].T = TP; } while(0)
	if(0 < best1 && 0 < best2) {
		if(paired) {
			flag |= 2; flag_r |= 2;
			if(0 < regT[0]) {
				flag |= 32; flag_r |= 16; o1 ^= 1;
				EMIT(0, 0, o1, best1, flag, regT, 0);
				EMIT(1, 1, o2, best2, flag_r, regT, nreg);
			} else {
				flag |= 16; flag_r |= 32; o2 ^= 1;
				for(int i = 0; i < nreg; ++i) regT[i] = -regT[i];
				EMIT(0, 1, o2, best2, flag_r, regT, 0);
				EMIT(1, 0, o1, best1, flag, regT, nreg);
			}
		} else {
			int s1 = best1, s2 = best2;
			if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
			else { flag |= 16; flag_r |= 32; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			if(0 < bT[0]) { o2 ^= 1; if(bT[nb2 - 1] < 0) s2 = -s2; }
			else { flag |= 32; flag_r |= 16; for(int i = 0; i < nb2; ++i) bT[i] = -bT[i]; }
			EMIT(0, 0, o1, s1, flag, regT, nreg);
			EMIT(1, 1, o2, s2, flag_r, bT, nb2);
		}
		ret = 0;
	} else if(best1) {
		int s1 = best1;
		flag |= 8; flag |= 32;
		if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
		else { flag |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
		EMIT(0, 0, o1, s1, flag, regT, nreg);
		ret = 2;
	} else if(best2) {
		int s2 = best2;
		flag_r |= 8; flag_r |= 32;
		if(0 < regT[0]) { o2 ^= 1; if(regT[nreg - 1] < 0) s2 = -s2; }
		else { flag_r |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
		EMIT(1, 1, o2, s2, flag_r, regT, nreg);
		ret = 1;
	}
	#undef EMIT
	free(buf);
	ws_free(w);
	return ret;
}
