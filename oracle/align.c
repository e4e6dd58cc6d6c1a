/* oracle/align.c -- TEST INFRASTRUCTURE (see kma_oracle.h).
 * CPU restatement of stage 3a for single-end reads:
 *   per-template position index   hashmapcci.c:95-199,470-505 (semantics only:
 *                                 k-mer -> ascending occurrence list, poly-A
 *                                 k-mer 0 never indexed :414-417)
 *   MEM seeding + stitching       KMA_score, align.c:509-748
 *   tails                         leadTailAln/trailTailAln, align.c:53-212
 *   chaining                      chainSeeds, chain.c:79-260
 *   global/semi-global DP         NW_score nw.c:642-890, NW_band_score :892-1188
 *   per-read filters              alnFragsSE alnfrags.c:1052-1218
 *   hit selection + accumulators  update_Scores updatescores.c:203-298
 */
#include "kma_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int tnuc(const uint64_t *s, int pos) { return (int) ((s[pos >> 5] << ((pos & 31) << 1)) >> 62); }

static inline uint64_t kmer_at(const uint64_t *seq, int pos, int k) {
	int ip = (pos & 31) << 1, w = pos >> 5, sh = 64 - (k << 1);
	if(ip <= sh) return (seq[w] << ip) >> sh;
	return ((seq[w] << ip) | (seq[w + 1] >> (64 - ip))) >> sh;
}

/* ---- per-template occurrence index ---------------------------------------- */
typedef struct { uint64_t key; int pos; } occ;
typedef struct { occ *o; int n; } tindex;

static int occ_cmp(const void *a, const void *b) {
	const occ *x = a, *y = b;
	if(x->key != y->key) return x->key < y->key ? -1 : 1;
	return x->pos - y->pos;
}

static void tindex_build(tindex *ix, const uint64_t *tseq, int tlen, int k) {
	int n = tlen - k + 1, c = 0;
	if(n < 0) n = 0;
	ix->o = malloc(sizeof(occ) * (size_t) (n ? n : 1));
	for(int i = 0; i < n; ++i) {
		uint64_t key = kmer_at(tseq, i, k);
		if(key == 0) continue;                 /* hashmapcci.c:414-417 */
		ix->o[c].key = key; ix->o[c].pos = i + 1; ++c;
	}
	qsort(ix->o, (size_t) c, sizeof(occ), occ_cmp);
	ix->n = c;
}

/* first index of key, *cnt = occurrences */
static int tindex_find(const tindex *ix, uint64_t key, int *cnt) {
	int lo = 0, hi = ix->n;
	while(lo < hi) { int mid = (lo + hi) >> 1; if(ix->o[mid].key < key) lo = mid + 1; else hi = mid; }
	int e = lo;
	while(e < ix->n && ix->o[e].key == key) ++e;
	*cnt = e - lo;
	return lo;
}

/* ---- DP workspace --------------------------------------------------------- */
typedef struct {
	int *D[2], *P[2]; long rowcap;
	uint8_t *E; long ecap;
	/* MEM arrays */
	int *tS, *tE, *qS, *qE, *w, *sc, *nx; int pcap, plen;
} aws;

static void aws_rows(aws *w, long need) {
	if(w->rowcap > need) return;
	long cap = need * 2 + 64;
	for(int i = 0; i < 2; ++i) { free(w->D[i]); free(w->P[i]); w->D[i] = calloc((size_t) cap, sizeof(int)); w->P[i] = calloc((size_t) cap, sizeof(int)); }
	w->rowcap = cap;
}
static void aws_E(aws *w, long need) {
	if(w->ecap > need) return;
	free(w->E); w->ecap = need * 2 + 64; w->E = calloc((size_t) w->ecap, 1);
}
static void aws_points(aws *w, int need) {
	if(w->pcap > need) return;
	int cap = need * 2 + 64;
	w->tS = realloc(w->tS, sizeof(int) * (size_t) cap); w->tE = realloc(w->tE, sizeof(int) * (size_t) cap);
	w->qS = realloc(w->qS, sizeof(int) * (size_t) cap); w->qE = realloc(w->qE, sizeof(int) * (size_t) cap);
	w->w = realloc(w->w, sizeof(int) * (size_t) cap); w->sc = realloc(w->sc, sizeof(int) * (size_t) cap);
	w->nx = realloc(w->nx, sizeof(int) * (size_t) cap);
	w->pcap = cap;
}

typedef struct { int score, len, pos, match, tGaps, qGaps; } aln;

/* call counters for tests: [0] full NW, [1] banded NW, [2] DP cells, [3] chain calls */
int64_t orc_counters[4] = {0, 0, 0, 0};

static aln degenerate(int t_len, int q_len, const orc_rewards *rw) {
	/* nw.c:662-684 */
	aln s = {0, 0, 0, 0, 0, 0};
	if(t_len == q_len) return s;
	if(t_len == 0) { s.len = q_len; s.tGaps = q_len; s.score = rw->W1 + (q_len - 1) * rw->U; }
	else { s.len = t_len; s.qGaps = t_len; s.score = rw->W1 + (t_len - 1) * rw->U; }
	return s;
}

/* walk the move matrix (nw.c:846-886): moves 1 diag, 2/3 gap in template
 * (query consumed), 4/5 gap in query (template consumed); a gap run ends on the
 * first cell carrying EITHER "may open" bit (16 | 32). stride = row pitch,
 * dn = column change per template step (0 full matrix, -1 banded). */
static void walk(const uint8_t *E, long stride, int m, int n, int dn, aln *s) {
	const uint8_t *row = E + (long) m * stride;
	s->len = s->match = s->tGaps = s->qGaps = 0;
	while(row[n] != 0) {
		int mv = row[n] & 7;
		if(mv == 1) {
			++s->match; row += stride; n += 1 + dn;
		} else if(mv >= 4) {
			while(!(row[n] >> 4)) { row += stride; n += dn; ++s->len; ++s->qGaps; }
			++s->qGaps; row += stride; n += dn;
		} else {
			while(!(row[n] >> 3)) { ++n; ++s->len; ++s->tGaps; }
			++s->tGaps; ++n;
		}
		++s->len;
	}
}

/* Frag_align of the traceback variants (nw.c:26-309, 310-640): aligned strings as codes 0-5 (5 = gap), the match
 * line ('|' / '_'), and how many query bases the walk left out at either end (Aln.start / Aln.end). pos = the
 * template length for circular joins (align.c:456). */
typedef struct { uint8_t *t, *s, *q; long cap; int start, end, pos; } trace;

static void trace_room(trace *tr, long need) {
	if(tr->cap > need) return;
	tr->cap = need * 2 + 64;
	tr->t = realloc(tr->t, (size_t) tr->cap); tr->s = realloc(tr->s, (size_t) tr->cap); tr->q = realloc(tr->q, (size_t) tr->cap);
}

/* the same walk, writing the columns (nw.c:256-305 full matrix, :586-635 band): q_pos = query index of column n */
static void walk_trace(const uint8_t *E, long stride, int m, int n, int dn, int q_pos, aln *s, trace *tr,
                       const uint64_t *tseq, int nuc_pos, int tlen_total, const uint8_t *query, int q_len) {
	const uint8_t *row = E + (long) m * stride;
	s->len = s->match = s->tGaps = s->qGaps = 0;
	while(row[n] != 0) {
		if(nuc_pos == tlen_total) nuc_pos = 0;
		const int mv = row[n] & 7;
		if(mv == 1) {
			tr->t[s->len] = (uint8_t) tnuc(tseq, nuc_pos); tr->q[s->len] = query[q_pos];
			tr->s[s->len] = tr->t[s->len] == tr->q[s->len] ? '|' : '_';
			++s->match; ++nuc_pos; row += stride; n += 1 + dn; ++q_pos;
		} else if(mv >= 4) {
			while(!(row[n] >> 4)) {
				tr->t[s->len] = (uint8_t) tnuc(tseq, nuc_pos); tr->q[s->len] = 5; tr->s[s->len] = '_';
				++nuc_pos; row += stride; n += dn; ++s->len; ++s->qGaps;
			}
			tr->t[s->len] = (uint8_t) tnuc(tseq, nuc_pos); tr->q[s->len] = 5; tr->s[s->len] = '_';
			++nuc_pos; row += stride; n += dn; ++s->qGaps;
		} else {
			while(!(row[n] >> 3)) {
				tr->t[s->len] = 5; tr->q[s->len] = query[q_pos]; tr->s[s->len] = '_';
				++n; ++q_pos; ++s->len; ++s->tGaps;
			}
			tr->t[s->len] = 5; tr->q[s->len] = query[q_pos]; tr->s[s->len] = '_';
			++n; ++q_pos; ++s->tGaps;
		}
		++s->len;
	}
	tr->end = q_len - q_pos;
}

/* nw.c:48-85: one side empty */
static aln degenerate_trace(int t_len, int q_len, const orc_rewards *rw, trace *tr, const uint64_t *tseq, int t_e, const uint8_t *query) {
	aln s = degenerate(t_len, q_len, rw);
	tr->start = tr->end = 0;
	if(t_len == q_len) return s;
	if(t_len == 0) {
		trace_room(tr, q_len + 2);
		memset(tr->s, '_', (size_t) q_len); memset(tr->t, 5, (size_t) q_len); memcpy(tr->q, query, (size_t) q_len);
	} else {
		trace_room(tr, t_len + 2);
		memset(tr->s, '_', (size_t) t_len); memset(tr->q, 5, (size_t) t_len);
		int nuc_pos = (t_e ? t_e : tr->pos) - 1;
		for(int m = t_len; m--;) {
			tr->t[m] = (uint8_t) tnuc(tseq, nuc_pos);
			if(--nuc_pos < 0) nuc_pos = tr->pos - 1;
		}
	}
	return s;
}

static aln nw_score(aws *w, const uint64_t *tseq, const uint8_t *qorg, int k, int t_s, int t_e, int q_s, int q_e,
                    const orc_rewards *rw, int tlen_total, trace *tr) {
	/* nw.c:642-890; with tr: NW, nw.c:26-309 (same fill, the walk writes the columns) */
	const int W1 = rw->W1, U = rw->U;
	int t_len = t_e - t_s, q_len = q_e - q_s;
	if(t_len < 0) t_len += tlen_total;
	const uint8_t *q = qorg + q_s;
	if(t_len == 0 || q_len == 0) return tr ? degenerate_trace(t_len, q_len, rw, tr, tseq, t_e, q) : degenerate(t_len, q_len, rw);
	orc_counters[0]++; orc_counters[2] += (int64_t) t_len * q_len;
	aws_rows(w, q_len + 2);
	aws_E(w, (long) (q_len + 2) * (t_len + 2));
	const long pitch = q_len + 1;
	const int low = (t_len + q_len) * (rw->MM + U + W1);
	int *Dc = w->D[0], *Dp = w->D[1], *Pc = w->P[0], *Pp = w->P[1];
	uint8_t *E = w->E, *Er = E + pitch * t_len; /* boundary row m = t_len */
	aln s; s.pos = 0; s.score = low;
	/* last column + last row by mode (nw.c:703-750) */
	for(int m = 0; m < t_len; ++m) E[pitch * m + q_len] = (0 < k) ? 0 : 5;
	if(!(0 < k)) E[pitch * (t_len - 1) + q_len] = 36;
	if(k == 2) {
		for(int n = q_len; n >= 0; --n) { Dp[n] = 0; Pp[n] = low; Er[n] = 0; }
	} else {
		for(int n = q_len - 1; n >= 0; --n) { Dp[n] = W1 + (q_len - 1 - n) * U; Pp[n] = low; Er[n] = 3; }
		Er[q_len - 1] = 18; Er[q_len] = 0; Dp[q_len] = 0; Pp[q_len] = 0;
	}
	int best_m = 0;
	int npos = t_e - 1;
	for(int m = t_len - 1; m >= 0; --m, --npos) {
		if(npos < 0) npos = tlen_total - 1;
		uint8_t *e = E + pitch * m;
		Dc[q_len] = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
		int Qprev = low;
		const int tn = tnuc(tseq, npos);
		for(int n = q_len - 1; n >= 0; --n) {
			uint8_t cell = 0, mv;
			int Q = Dc[n + 1] + W1;
			int P = Dp[n] + W1;
			int D;
			if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
			int x = Qprev + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
			x = Pp[n] + U;
			if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
			x = Dp[n + 1] + rw->d[tn][q[n]];
			if(D <= x) { D = x; cell |= 1; } else cell |= mv;
			Dc[n] = D; Pc[n] = P; e[n] = cell; Qprev = Q;
		}
		if(k < 0 && s.score < Dc[0]) { s.score = Dc[0]; best_m = m; }
		int *t = Dc; Dc = Dp; Dp = t; t = Pc; Pc = Pp; Pp = t;
	}
	int sm = 0, sn = 0;
	if(k < 0) {
		sm = best_m;
		if(k == -2) {
			for(int n = 0; n < q_len; ++n) if(s.score <= Dp[n]) { s.score = Dp[n]; sm = 0; sn = n; }
		}
	} else {
		s.score = Dp[0];
	}
	if(tr) {
		trace_room(tr, (long) t_len + q_len + 2);
		tr->start = sn;
		walk_trace(E, pitch, sm, sn, 0, sn, &s, tr, tseq, sm + t_s, tlen_total, q, q_len);
	} else walk(E, pitch, sm, sn, 0, &s);
	return s;
}

static aln nw_band_score(aws *w, const uint64_t *tseq, const uint8_t *qorg, int k, int t_s, int t_e, int q_s, int q_e,
                         int band, const orc_rewards *rw, int tlen_total, trace *tr) {
	/* nw.c:892-1188. Band columns are indexed relative to a centre diagonal
	 * that moves one query position per template row; column n of row m is
	 * column n-1 of row m+1. */
	const int W1 = rw->W1, U = rw->U;
	int t_len = t_e - t_s, q_len = q_e - q_s;
	if(t_len < 0) t_len += tlen_total;
	const uint8_t *q = qorg + q_s;
	if(t_len == 0 || q_len == 0) return tr ? degenerate_trace(t_len, q_len, rw, tr, tseq, t_e, q) : degenerate(t_len, q_len, rw);
	if(band & 1) ++band;
	orc_counters[1]++; orc_counters[2] += (int64_t) t_len * (band + 1);
	const int half = band >> 1, bq = band + 1;
	aws_rows(w, (long) band * 2 + 8);
	aws_E(w, (long) (band + 3) * (t_len + 2));
	const long pitch = bq + 1;
	const int low = (t_len + q_len) * (rw->MM + U + W1);
	int *Dc = w->D[0], *Dp = w->D[1], *Pc = w->P[0], *Pp = w->P[1];
	uint8_t *E = w->E, *Er = E + pitch * t_len;
	aln s; s.pos = 0; s.score = low;
	int c = (t_len + q_len) >> 1;
	int sn = q_len - 1 - (c - half);
	if(k != 2) {
		for(int n = sn - 1; n >= 0; --n) { Dp[n] = W1 + (sn - n - 1) * U; Pp[n] = low; Er[n] = 3; }
		Er[sn - 1] = 18; Er[sn] = 0; Dp[sn] = 0; Pp[sn] = 0;
	} else {
		for(int n = sn; n >= 0; --n) { Dp[n] = 0; Pp[n] = low; Er[n] = 0; }
	}
	int bm = 0, bn = 0, en = 0, n = 0;
	int npos = t_e - 1;
	for(int m = t_len - 1; m >= 0; --m, --npos, --c) {
		if(npos < 0) npos = tlen_total - 1;
		uint8_t *e = E + pitch * m;
		int sq = c + half, eq = c - half;
		if(eq < 0) { eq = 0; ++en; } else en = 0;
		int Qprev = low;
		if(sq < q_len - 1) {
			sn = bq - 1; Dc[bq] = low; e[bq] = 37;
		} else {
			sq = q_len - 1; sn = en + (q_len - eq);
			Dc[sn] = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
			e[sn] = (0 < k) ? 0 : 37;
			--sn;
		}
		const int tn = tnuc(tseq, npos);
		int qp = sq;
		for(n = sn; n > en; --qp, --n) {
			uint8_t cell = 0, mv;
			int Q = Dc[n + 1] + W1;
			int P = Dp[n - 1] + W1;
			int D;
			if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
			int x = Qprev + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
			x = Pp[n - 1] + U;
			if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
			x = Dp[n] + rw->d[tn][q[qp]];
			if(D <= x) { D = x; cell |= 1; } else cell |= mv;
			Dc[n] = D; Pc[n] = P; e[n] = cell; Qprev = Q;
		}
		/* band edge: no gap-in-query state (nw.c:1079-1105) */
		{
			uint8_t cell = 0, mv;
			int Q = Dc[n + 1] + W1, x = Qprev + U;
			if(Q < x) { Q = x; mv = 3; } else { mv = 2; cell |= 16; }
			Pc[n] = low;
			int D = Dp[n] + rw->d[tn][q[qp]];
			if(Q <= D) cell |= 1; else { D = Q; cell |= mv; }
			Dc[n] = D; e[n] = cell;
		}
		if(eq == 0 && k < 0 && s.score < Dc[n]) { s.score = Dc[n]; bm = m; bn = n; }
		int *t = Dc; Dc = Dp; Dp = t; t = Pc; Pc = Pp; Pp = t;
	}
	int q_pos = 0;
	if(bm == 0) { bn = en; s.score = Dp[en]; }
	if(k == -2) {
		for(n = en; n < bq; ++n) if(s.score <= Dp[n]) { s.score = Dp[n]; bm = 0; bn = n; q_pos = n - en; }
	}
	if(tr) {
		trace_room(tr, (long) t_len + q_len + 2);
		tr->start = q_pos;
		walk_trace(E, pitch, bm, bn, -1, q_pos, &s, tr, tseq, bm + t_s, tlen_total, q, q_len);
	} else walk(E, pitch, bm, bn, -1, &s);
	return s;
}

/* ---- chaining, chain.c:79-260 ------------------------------------------- */
static int mism_score(int span, int k, const orc_rewards *rw) {
	/* heuristic substitution model shared by the end / link / start terms */
	int Ms, MMs;
	if(span == 2) { MMs = 2; Ms = 0; }
	else {
		MMs = span / k + (span % k ? 1 : 0);
		if(MMs < 2) MMs = 2;
		Ms = span - MMs; if(k < Ms) Ms = k; if(MMs < Ms) Ms = MMs;
	}
	return Ms * rw->M + MMs * rw->MM;
}

static int chain_seeds(aws *w, int q_len, int t_len, int k, const orc_rewards *rw, unsigned *mapQ) {
	const int W1 = rw->W1, U = rw->U, M = rw->M;
	const int n = w->plen;
	int best = 0, second = 0, bestPos = n - 1;
	w->sc[n] = 0; w->nx[n] = 0;
	for(int i = n - 1; i >= 0; --i) {
		const int weight = w->w[i] * M, tEnd = w->tE[i], qEnd = w->qE[i];
		w->nx[i] = 0;
		int span = (t_len - tEnd < q_len - qEnd) ? t_len - tEnd : q_len - qEnd;
		int gap = span - 1;
		gap = gap ? gap * U + W1 : W1;                  /* chain.c:104-111 (negative falls in the first arm) */
		int sub = mism_score(span, k, rw);
		int score = weight + (sub < gap ? gap : sub);
		const int lim = (n < i + 128) ? n : i + 128;
		for(int j = i + 1; j < lim; ++j) {
			if(qEnd < w->qS[j]) {
				if(tEnd < w->tS[j]) {
					const int tGap = w->tS[j] - tEnd, qGap = w->qS[j] - qEnd;
					int g = abs(tGap - qGap);
					if(g) g = (g - 1) * U + W1;
					g += weight + w->sc[j] + mism_score(tGap < qGap ? tGap : qGap, k, rw);
					if(score <= g) { score = g; w->nx[i] = j; }
				} else if(k <= w->tE[j] - tEnd) {
					int g = w->qS[j] - qEnd;
					if(g) g = (g - 1) * U + W1;
					g += weight + w->sc[j] - (w->tS[j] - tEnd) * M;
					if(score < g) { score = g; w->nx[i] = j; }
				}
			} else if(k <= w->qE[j] - qEnd) {
				const int tStart = w->tS[j] + qEnd - w->qS[j];
				if(tEnd < tStart) {
					int g = tStart - tEnd;
					if(g) g = (g - 1) * U + W1;
					g += weight + w->sc[j] - (tStart - tEnd) * M;
					if(score < g) { score = g; w->nx[i] = j; }
				}
			}
		}
		if(w->nx[i]) w->w[i] += w->w[w->nx[i]] - k + 1; else w->w[i] -= k - 1;
		w->sc[i] = score;
		span = (w->tS[i] < w->qS[i]) ? w->tS[i] : w->qS[i];
		gap = span - 1;
		if(0 < gap) gap = gap * U + W1; else if(gap == 0) gap = W1; else gap = 0;
		sub = mism_score(span, k, rw);
		score += sub < gap ? gap : sub;
		if(best <= score) {
			if(w->nx[i] != bestPos) second = best;
			best = score; bestPos = i;
		} else if(second <= score && w->nx[i] != bestPos) {
			second = best;
		}
	}
	if(0 < best) {
		double wq = w->w[bestPos] / 10.0; if(1 < wq) wq = 1;
		*mapQ = (unsigned) ceil(40 * (1 - 1.0 * second / best) * wq * log(best));
	} else *mapQ = 0;
	w->sc[bestPos] = best;
	return bestPos;
}

/* ---- tails, align.c:53-212 ------------------------------------------------ */
static aln lead_tail(aws *w, const uint64_t *tseq, const uint8_t *qseq, int t_e, int t_len, int q_e, int bw, const orc_rewards *rw) {
	aln s = {0, 0, t_e, 0, 0, 0};
	if(!q_e) return s;
	int t_s = 0, q_s = 0;
	if((q_e << 1) < t_e || (q_e + bw) < t_e) t_s = t_e - (q_e + (q_e < bw ? q_e : bw));
	else if((t_e << 1) < q_e || (t_e + bw) < q_e) q_s = q_e - (t_e + (t_e < bw ? t_e : bw));
	if(t_e - t_s > 0 && q_e - q_s > 0) {
		const int band = abs(t_e - t_s - q_e + q_s) + bw;
		const int mode = -1 - (t_s == 0);
		aln r;
		if(q_e - q_s <= band || t_e - t_s <= band) r = nw_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, rw, t_len, NULL);
		else r = nw_band_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, band, rw, t_len, NULL);
		s.pos -= r.len - r.tGaps;
		s.score = r.score; s.len = r.len; s.match = r.match; s.tGaps = r.tGaps; s.qGaps = r.qGaps;
	}
	return s;
}

static void trail_tail(aws *w, aln *s, const uint64_t *tseq, const uint8_t *qseq, int t_s, int t_len, int q_s, int q_len, int bw, const orc_rewards *rw) {
	int q_e = q_len, t_e = t_len;
	if(((q_len - q_s) << 1) < (t_len - t_s) || (q_len - q_s + bw) < (t_len - t_s)) {
		t_e = q_len - q_s; t_e = t_s + (t_e + (t_e < bw ? t_e : bw));
	} else if(((t_len - t_s) << 1) < (q_len - q_s) || (t_len - t_s + bw) < (q_len - q_s)) {
		q_e = t_len - t_s; q_e = q_s + (q_e + (q_e < bw ? q_e : bw));
	}
	if(t_e - t_s > 0 && q_e - q_s > 0) {
		const int band = abs(t_e - t_s - q_e + q_s) + bw;
		const int mode = 1 + (t_e == t_len);
		aln r;
		if(q_e - q_s <= band || t_e - t_s <= band) r = nw_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, rw, t_len, NULL);
		else r = nw_band_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, band, rw, t_len, NULL);
		s->score += r.score; s->len += r.len; s->match += r.match; s->tGaps += r.tGaps; s->qGaps += r.qGaps;
	}
}

/* ---- KMA_score, align.c:509-748 ------------------------------------------ */
static void add_mem(aws *w, const uint64_t *tseq, int t_len, const uint8_t *qseq, int j, int pos1, int k, int segstop, int *qend_out) {
	aws_points(w, w->plen + 2);
	int prev = pos1 - 2, kk;
	for(kk = j - 1; 0 <= kk && 0 <= prev && qseq[kk] == tnuc(tseq, prev); --kk) --prev;
	const int m = w->plen;
	w->qS[m] = kk + 1; w->tS[m] = prev + 2;
	int value = pos1 + k - 1, l = j + k;
	while(l < segstop && value < t_len && qseq[l] == tnuc(tseq, value)) { ++l; ++value; }
	w->qE[m] = l; w->tE[m] = value + 1;
	w->w[m] = w->qE[m] - w->qS[m];
	w->plen = m + 1;
	*qend_out = l;
}

static const aln FAIL = {0, 1, 0, 0, 0, 0};

static aln kma_score(aws *w, const tindex *ix, const uint64_t *tseq, int t_len, int k, const uint8_t *qseq, int q_len,
                     int q_start, int q_end, const uint64_t *qcomp, const int *N /* N[0] = count incl. sentinel */,
                     int mq, const orc_rewards *rw, int preseeded) {
	const int bw = 64;
	/* align.c:530-532: MEMs left in `points` by anker_rc_comp are used as they are */
	const int use_pre = preseeded && w->plen > 0;
	if(!use_pre) w->plen = 0;
	int j = q_start;
	for(int i = 1; !use_pre && i <= N[0]; ++i) {
		int end = (i != N[0]) ? N[i] - k + 1 : q_end - k + 1;
		while(j < end) {
			int cnt;
			const uint64_t key = kmer_at(qcomp, j, k);
			const int first = key ? tindex_find(ix, key, &cnt) : (cnt = 0, 0);
			if(cnt == 0) { ++j; continue; }
			const int segstop = end + k - 1;
			if(cnt == 1) {
				int qe;
				add_mem(w, tseq, t_len, qseq, j, ix->o[first].pos, k, segstop, &qe);
				j = qe;
			} else {
				int bias = j;
				for(int c = 0; c < cnt; ++c) {
					int qe;
					add_mem(w, tseq, t_len, qseq, j, ix->o[first + c].pos, k, segstop, &qe);
					if(bias < qe) bias = qe;
				}
				j = bias + 1;
			}
		}
		j = N[i] + 1;
	}
	if(!w->plen) return FAIL;
	aws_points(w, w->plen + 2);
	unsigned mapQ = 0;
	int start = chain_seeds(w, q_len, t_len, k, rw, &mapQ);
	if(mapQ < (unsigned) mq || w->sc[start] < k) return FAIL;
	aln S = lead_tail(w, tseq, qseq, w->tS[start] - 1, t_len, w->qS[start], bw, rw);
	for(;;) {
		const int span = w->qE[start] - w->qS[start];
		S.len += span; S.match += span;
		for(int i = w->qS[start]; i < w->qE[start]; ++i) S.score += rw->d[qseq[i]][qseq[i]];
		if(!w->nx[start]) break;
		int q_s = w->qE[start], t_s = w->tE[start] - 1, t_l;
		start = w->nx[start];
		if(w->qS[start] < q_s) { w->tS[start] += q_s - w->qS[start]; w->qS[start] = q_s; }
		int t_e = w->tS[start] - 1;
		if(t_e < t_s) {
			if(t_s <= w->tE[start]) { w->qS[start] += t_s - t_e; t_e = t_s; t_l = 0; }
			else t_l = t_len - t_s + t_e;
		} else t_l = t_e - t_s;
		const int q_e = w->qS[start];
		if(abs(t_l - q_e + q_s) * rw->U > q_len * rw->M || t_l > q_len || q_e - q_s > (q_len >> 1)) return FAIL;
		if(t_l > 0 || q_e - q_s > 0) {
			const int band = abs(t_l - q_e + q_s) + bw;
			aln r;
			if(q_e - q_s <= band || t_l <= band) r = nw_score(w, tseq, qseq, 0, t_s, t_e, q_s, q_e, rw, t_len, NULL);
			else r = nw_band_score(w, tseq, qseq, 0, t_s, t_e, q_s, q_e, band, rw, t_len, NULL);
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
	}
	trail_tail(w, &S, tseq, qseq, w->tE[start] - 1, t_len, w->qE[start], q_len, bw, rw);
	return S;
}

/* ---- KMA, align.c:214-507: the stage-3c aligner. Same chain and joins as KMA_score, but (i) the seeding loop is the
 * byte-wise one (a stretch must be LONGER than k to be probed, align.c:258,308,364), (ii) NW / NW_band write the aligned
 * columns, (iii) gaps at the very start / end of the template are trimmed off the tails (align.c:97-112, 180-196).
 * `al` receives the columns (codes 0-5), the number of columns is the returned len; al->start / al->end = read bases left
 * unaligned before / after (soft clips of the SAM record). trimSeeds (chain.c:496) is a no-op at its default ts = 0. */
static void al_put(trace *al, long at, const uint8_t *t, const uint8_t *ss, const uint8_t *q, long n) {
	trace_room(al, at + n + 2);
	memcpy(al->t + at, t, (size_t) n); memcpy(al->s + at, ss, (size_t) n); memcpy(al->q + at, q, (size_t) n);
}

static aln kma_trace(aws *w, const tindex *ix, const uint64_t *tseq, int t_len, int k, const uint8_t *qseq, int q_len,
                     int q_start, int q_end, int mq, const orc_rewards *rw, trace *al, trace *fr, unsigned *mapQ_out, int preseeded) {
	const int bw = 64;
	const uint64_t mask = (k < 32) ? ((1ull << (2 * k)) - 1) : ~0ull;
	al->start = al->end = 0;
	/* MEMs left by anker_rc are used as they are (align.c:245-247) */
	if(!preseeded) w->plen = 0;
	int i = preseeded ? q_end : q_start;
	while(i < q_end) {
		int end = -1;
		for(int x = i; x < q_len; ++x) if(qseq[x] == 4) { end = x; break; }     /* charpos(qseq, 4, i, q_len) */
		if(end == -1) end = q_end;
		uint64_t key = 0;
#define RESTART_KEY() do { if(i < end - k) { key = 0; for(int x = 0; x < k - 1; ++x) key = (key << 2) | qseq[i + x]; i += k - 1; } else i = end + 1; } while(0)
		RESTART_KEY();
		while(i < end) {
			key = ((key << 2) | qseq[i]) & mask;
			int cnt = 0;
			const int first = key ? tindex_find(ix, key, &cnt) : 0;
			if(cnt == 0) { ++i; continue; }
			i -= k - 1;
			if(cnt == 1) {
				int qe;
				add_mem(w, tseq, t_len, qseq, i, ix->o[first].pos, k, end, &qe);
				i = qe;
			} else {
				int bias = i;
				for(int c = 0; c < cnt; ++c) {
					int qe;
					add_mem(w, tseq, t_len, qseq, i, ix->o[first + c].pos, k, end, &qe);
					if(bias < qe) bias = qe;
				}
				i = bias + 1;
			}
			RESTART_KEY();
		}
#undef RESTART_KEY
		i = end + 1;
	}
	if(!w->plen) return FAIL;
	aws_points(w, w->plen + 2);
	unsigned mapQ = 0;
	int start = chain_seeds(w, q_len, t_len, k, rw, &mapQ);
	*mapQ_out = mapQ;
	if(mapQ < (unsigned) mq || w->sc[start] < k) return FAIL;
	trace_room(al, (long) 2 * (q_len + t_len) + 64);

	/* leading tail, leadTailAln with Frag_align (align.c:53-131) */
	aln S = {0, 0, w->tS[start] - 1, 0, 0, 0};
	{
		const int t_e = w->tS[start] - 1, q_e = w->qS[start];
		if(q_e) {
			int t_s = 0, q_s = 0;
			if((q_e << 1) < t_e || (q_e + bw) < t_e) t_s = t_e - (q_e + (q_e < bw ? q_e : bw));
			else if((t_e << 1) < q_e || (t_e + bw) < q_e) q_s = q_e - (t_e + (t_e < bw ? t_e : bw));
			if(t_e - t_s > 0 && q_e - q_s > 0) {
				const int band = abs(t_e - t_s - q_e + q_s) + bw, mode = -1 - (t_s == 0);
				aln r;
				fr->start = fr->end = 0;
				if(q_e - q_s <= band || t_e - t_s <= band) r = nw_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, rw, t_len, fr);
				else r = nw_band_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, band, rw, t_len, fr);
				int bias = 0;
				if(t_s == 0) {
					while(bias < r.len && (fr->t[bias] == 5 || fr->q[bias] == 5)) {
						if(fr->t[bias] == 5) { --r.tGaps; ++fr->start; } else --r.qGaps;
						++bias;
					}
					r.len -= bias;
				}
				al_put(al, 0, fr->t + bias, fr->s + bias, fr->q + bias, r.len);
				al->start = q_s + fr->start;
				S.pos -= r.len - r.tGaps;
				S.score = r.score; S.len = r.len; S.match = r.match; S.tGaps = r.tGaps; S.qGaps = r.qGaps;
			} else al->start = q_s;
		}
	}
	for(;;) {
		const int span = w->qE[start] - w->qS[start];
		trace_room(al, (long) S.len + span + 2);
		memcpy(al->t + S.len, qseq + w->qS[start], (size_t) span); memset(al->s + S.len, '|', (size_t) span);
		memcpy(al->q + S.len, qseq + w->qS[start], (size_t) span);
		S.len += span; S.match += span;
		for(int x = w->qS[start]; x < w->qE[start]; ++x) S.score += rw->d[qseq[x]][qseq[x]];
		if(!w->nx[start]) break;
		int q_s = w->qE[start], t_s = w->tE[start] - 1, t_l;
		start = w->nx[start];
		if(w->qS[start] < q_s) { w->tS[start] += q_s - w->qS[start]; w->qS[start] = q_s; }
		int t_e = w->tS[start] - 1;
		if(t_e < t_s) {
			if(t_s <= w->tE[start]) { w->qS[start] += t_s - t_e; t_e = t_s; t_l = 0; }
			else { fr->pos = t_len; t_l = t_len - t_s + t_e; }
		} else t_l = t_e - t_s;
		const int q_e = w->qS[start];
		if(abs(t_l - q_e + q_s) * rw->U > q_len * rw->M || t_l > q_len || q_e - q_s > (q_len >> 1)) return FAIL;
		if(t_l > 0 || q_e - q_s > 0) {
			const int band = abs(t_l - q_e + q_s) + bw;
			aln r;
			if(q_e - q_s <= band || t_l <= band) r = nw_score(w, tseq, qseq, 0, t_s, t_e, q_s, q_e, rw, t_len, fr);
			else r = nw_band_score(w, tseq, qseq, 0, t_s, t_e, q_s, q_e, band, rw, t_len, fr);
			al_put(al, S.len, fr->t, fr->s, fr->q, r.len);
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
	}
	{	/* trailing tail, trailTailAln with Frag_align (align.c:140-212) */
		const int t_s = w->tE[start] - 1, q_s = w->qE[start];
		int q_e = q_len, t_e = t_len;
		if(((q_len - q_s) << 1) < (t_len - t_s) || (q_len - q_s + bw) < (t_len - t_s)) {
			t_e = q_len - q_s; t_e = t_s + (t_e + (t_e < bw ? t_e : bw));
		} else if(((t_len - t_s) << 1) < (q_len - q_s) || (t_len - t_s + bw) < (q_len - q_s)) {
			q_e = t_len - t_s; q_e = q_s + (q_e + (q_e < bw ? q_e : bw));
		}
		fr->end = 0;
		if(t_e - t_s > 0 && q_e - q_s > 0) {
			const int band = abs(t_e - t_s - q_e + q_s) + bw, mode = 1 + (t_e == t_len);
			aln r;
			if(q_e - q_s <= band || t_e - t_s <= band) r = nw_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, rw, t_len, fr);
			else r = nw_band_score(w, tseq, qseq, mode, t_s, t_e, q_s, q_e, band, rw, t_len, fr);
			if(t_e == t_len) {
				int bias = r.len - 1;
				while(bias && (fr->t[bias] == 5 || fr->q[bias] == 5)) {
					if(fr->t[bias] == 5) { --r.tGaps; ++fr->end; } else --r.qGaps;
					--bias;
				}
				++bias;
				if(bias != r.len) r.len = bias;
			}
			al_put(al, S.len, fr->t, fr->s, fr->q, r.len);
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
		al->end = q_len - q_e + fr->end;
	}
	return S;
}

/* ---- anker_rc_comp, align.c:993-1176 (strand decision of a stage-2 strand tie) ---- */
static int tindex_has(const tindex *ix, uint64_t key) { int c; if(!key) return 0; tindex_find(ix, key, &c); return c > 0; }

static int anker_rc_comp(aws *w, const tindex *ix, const uint64_t *tseq, int t_len, int k,
                         const uint8_t *qseq_f, const uint8_t *qseq_r, int q_len,
                         const uint64_t *comp_f, const int *N_f, const uint64_t *comp_r, const int *N_r, int one2one) {
	int bestScore = 0, score = 0, score_r = 0, mem_count = 0, tot = 0, plen = 0;
	for(int rc = 0; rc < 2; ++rc) {
		const uint8_t *qseq = rc ? qseq_r : qseq_f;
		const uint64_t *seq = rc ? comp_r : comp_f;
		const int *N = rc ? N_r : N_f;       /* N[0] = count incl. the q_len sentinel */
		int i = 0;
		if(rc) { score = score_r; plen = mem_count; }
		else {
			/* preseed, align.c:750-768: every k-th k-mer built from the byte codes; bytes past the read
			 * end are whatever the reference's buffer held -- taken as 0 here */
			int hit = 0;
			for(i = 0; i < q_len && !hit; i += k) {
				uint64_t key = 0;
				for(int x = 0; x < k; ++x) key = (x ? (key << 2) : 0) | (uint64_t) ((i + x < q_len) ? qseq[i + x] : 0);
				if(tindex_has(ix, key)) hit = 1;
			}
			i = hit ? 0 : i;
		}
		score_r = 0; mem_count = 0;
		int ni = 0;
		while(i < q_len) {
			const int end = N[++ni] - k + 1;
			while(i < end) {
				int cnt;
				const uint64_t key = kmer_at(seq, i, k);
				const int first = key ? tindex_find(ix, key, &cnt) : (cnt = 0, 0);
				if(cnt == 0) { ++i; continue; }
				if(cnt == 1) {
					aws_points(w, tot + 2);
					int value = ix->o[first].pos, prev = value - 2, j;
					for(j = i - 1; 0 <= j && 0 <= prev && qseq[j] == tnuc(tseq, prev); --j) { --prev; ++score_r; }
					w->qS[tot] = j + 1; w->tS[tot] = prev + 2;
					value += k - 1; i += k; score_r += k;
					while(i < end && value < t_len && qseq[i] == tnuc(tseq, value)) { ++i; ++value; ++score_r; }
					w->qE[tot] = i; w->tE[tot] = value + 1;
					w->w[tot] = w->tE[tot] - w->tS[tot];
					++mem_count; ++tot;
					++i;
				} else {
					score_r += k;
					int bias = i;
					for(int c = 0; c < cnt; ++c) {
						aws_points(w, tot + 2);
						int value = ix->o[first + c].pos, prev = value - 2, j, kk = i;
						for(j = kk - 1; 0 <= j && 0 <= prev && qseq[j] == tnuc(tseq, prev); --j) --prev;
						w->qS[tot] = j + 1; w->tS[tot] = prev + 2;
						value += k - 1; kk += k;
						while(kk < end && value < t_len && qseq[kk] == tnuc(tseq, value)) { ++kk; ++value; }
						w->qE[tot] = kk; w->tE[tot] = value + 1;
						w->w[tot] = w->qE[tot] - w->qS[tot];
						++mem_count; ++tot;
						if(bias < kk) bias = kk;
					}
					score_r += bias - i;
					i = bias + 1;
				}
			}
			i = end + k;
		}
		if(bestScore < score_r) bestScore = score_r;
	}
	if(one2one && bestScore < k && bestScore * k < (q_len - k - bestScore)) { w->plen = 0; return 0; }
	if(bestScore == score) { w->plen = plen; return bestScore; }
	if(plen) {
		for(int x = 0; x < mem_count; ++x) {
			w->tS[x] = w->tS[plen + x]; w->tE[x] = w->tE[plen + x]; w->qS[x] = w->qS[plen + x];
			w->qE[x] = w->qE[plen + x]; w->w[x] = w->w[plen + x];
		}
	}
	w->plen = mem_count;
	return -bestScore;
}

/* ---- alnFragsSE + update_Scores ----------------------------------------- */
struct orc_aligner {
	const orc_db *db;
	tindex *ix;      /* lazily built, one per template */
	aws w;
	uint8_t *q, *qr; uint64_t *rc; int *Nf, *Nr; int qcap, ncap;
	int *bt, *bs, *be, *bsc, *bl; int hcap;
};

orc_aligner *orc_aligner_new(const orc_db *db) {
	orc_aligner *a = calloc(1, sizeof *a);
	a->db = db;
	a->ix = calloc(db->DB_size + 1, sizeof(tindex));
	return a;
}

void orc_aligner_free(orc_aligner *a) {
	if(!a) return;
	for(uint32_t i = 0; i <= a->db->DB_size; ++i) free(a->ix[i].o);
	free(a->ix);
	for(int i = 0; i < 2; ++i) { free(a->w.D[i]); free(a->w.P[i]); }
	free(a->w.E); free(a->w.tS); free(a->w.tE); free(a->w.qS); free(a->w.qE); free(a->w.w); free(a->w.sc); free(a->w.nx);
	free(a->q); free(a->qr); free(a->rc); free(a->Nf); free(a->Nr);
	free(a->bt); free(a->bs); free(a->be); free(a->bsc); free(a->bl);
	free(a);
}

static void unpack_bytes(const uint64_t *seq, int len, const int *N, uint8_t *out) {
	for(int i = 0; i < len; ++i) out[i] = (uint8_t) tnuc(seq, i);
	for(int i = 1; i <= N[0]; ++i) out[N[i]] = 4;
	out[len] = 0;
}

int orc_align_se(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap,
                 const uint64_t *seq, int seqlen, const int *N, int nN,
                 int rc_flag, int flag, const int *T, int nT,
                 int *n_hits, int *best_score, int *out_flag, int *ht, int *hs, int *he, int *hscore,
                 uint64_t *alignment_scores, uint64_t *uniq_alignment_scores) {
	const orc_db *db = a->db;
	const int k = db->kmersize, q_len = seqlen;
	*n_hits = 0; *best_score = 0; *out_flag = flag;
	if(nT == 0 || q_len < k) return 0;
	const int words = (seqlen + 31) >> 5;
	if(a->qcap < seqlen + 2) {
		a->qcap = 2 * seqlen + 66;
		a->q = realloc(a->q, (size_t) a->qcap); a->qr = realloc(a->qr, (size_t) a->qcap);
		a->rc = realloc(a->rc, sizeof(uint64_t) * (size_t) ((a->qcap >> 5) + 4));
	}
	if(a->ncap < nN + 3) { a->ncap = 2 * nN + 66; a->Nf = realloc(a->Nf, sizeof(int) * (size_t) a->ncap); a->Nr = realloc(a->Nr, sizeof(int) * (size_t) a->ncap); }
	if(a->hcap < nT + 1) {
		a->hcap = 2 * nT + 16;
		a->bt = realloc(a->bt, sizeof(int) * (size_t) a->hcap); a->bs = realloc(a->bs, sizeof(int) * (size_t) a->hcap);
		a->be = realloc(a->be, sizeof(int) * (size_t) a->hcap); a->bsc = realloc(a->bsc, sizeof(int) * (size_t) a->hcap);
		a->bl = realloc(a->bl, sizeof(int) * (size_t) a->hcap);
	}
	a->Nf[0] = nN; memcpy(a->Nf + 1, N, sizeof(int) * (size_t) nN);
	/* the S2 stream carries the reverse-complemented read when flag & 16 (savekmers.c:3049) */
	const uint64_t *qcomp = seq; int *Nq = a->Nf;
	if(flag & 16) {
		orc_rc(seq, seqlen, a->Nf, a->rc, a->Nr);
		a->rc[words] = 0;
		qcomp = a->rc; Nq = a->Nr;
	}
	unpack_bytes(qcomp, seqlen, Nq, a->q);
	if(rc_flag < 0) {
		/* both orientations are needed (alnfrags.c:1061-1069); flag is 0 for ties so qcomp == seq */
		orc_rc(seq, seqlen, a->Nf, a->rc, a->Nr);
		a->rc[words] = 0;
		unpack_bytes(a->rc, seqlen, a->Nr, a->qr);
		a->Nr[0] += 1; a->Nr[a->Nr[0]] = q_len;
	}
	Nq[0] += 1; Nq[Nq[0]] = q_len;        /* sentinel, alnfrags.c:1071-1072 */

	double bestScore = 0; int bestRead = 0, hits = 0;
	for(int ti = 0; ti < nT; ++ti) {
		const int tmpl = T[ti], at = abs(tmpl);
		const int t_len = db->tlen[at];
		const uint64_t *tseq = db->tseq + db->tseq_off[at];
		if(!a->ix[at].o) tindex_build(&a->ix[at], tseq, t_len, k);
		aln st;
		int tmpl_out = tmpl;
		if(rc_flag < 0) {
			/* strand tie: decide per template by MEM coverage (alnfrags.c:1101-1124) */
			const int rcv = anker_rc_comp(&a->w, &a->ix[at], tseq, t_len, k, a->q, a->qr, q_len, qcomp, Nq, a->rc, a->Nr, 1);
			if(rcv < 0) { tmpl_out = -at; st = kma_score(&a->w, &a->ix[at], tseq, t_len, k, a->qr, q_len, 0, q_len, a->rc, a->Nr, ap->mq, rw, 1); }
			else if(rcv) { tmpl_out = at; st = kma_score(&a->w, &a->ix[at], tseq, t_len, k, a->q, q_len, 0, q_len, qcomp, Nq, ap->mq, rw, 1); }
			else { st.score = 0; st.pos = 0; st.len = 0; st.match = 0; st.tGaps = 0; st.qGaps = 0; a->w.plen = 0; }
		} else {
			st = kma_score(&a->w, &a->ix[at], tseq, t_len, k, a->q, q_len, 0, q_len, qcomp, Nq, ap->mq, rw, 0);
		}
		const int aln_len = st.len, start = st.pos;
		int end = start + aln_len - st.tGaps;
		if(t_len < end) end -= t_len;
		double denom;
		if(q_len <= aln_len || t_len <= aln_len) denom = aln_len; else denom = q_len < t_len ? q_len : t_len;
		int read_score = st.score; double score;
		if(ap->minlen <= aln_len && ((ap->mrc * q_len <= st.len - st.qGaps) || (ap->mrc * t_len <= st.len - st.tGaps))) score = read_score / denom;
		else { read_score = 0; score = 0; }
		if(k < read_score && ap->scoreT <= score) {
			a->bt[hits] = tmpl_out; a->bs[hits] = start; a->be[hits] = end; a->bsc[hits] = read_score; a->bl[hits] = aln_len; ++hits;
			if(bestScore < score) bestScore = score;
			if(bestRead < read_score) bestRead = read_score;
		}
	}
	Nq[0] -= 1;
	if(!(bestRead > k)) { *out_flag = flag | 4; return 0; }
	/* update_Scores, minFrac == 1.0 branch (updatescores.c:217-234) */
	int c = 0;
	for(int i = 0; i < hits; ++i) {
		const double ms = a->bsc[i] / a->bl[i];       /* integer division, then widened */
		if(ms == bestScore || a->bsc[i] == bestRead) {
			ht[c] = a->bt[i]; hs[c] = a->bs[i]; he[c] = a->be[i]; hscore[c] = a->bsc[i]; ++c;
			if(alignment_scores) alignment_scores[abs(a->bt[i])] += (uint64_t) a->bsc[i];
		}
	}
	if(c == 1 && uniq_alignment_scores) uniq_alignment_scores[abs(ht[0])] += (uint64_t) bestRead;
	*n_hits = c; *best_score = bestRead;
	return c;
}

/* function-level taps for kernel tests */
void orc_nw_tap(const uint64_t *tseq, int tlen_total, const uint8_t *q, int k, int t_s, int t_e, int q_s, int q_e,
                int band /* < 0: full matrix */, const orc_rewards *rw, int out[6]) {
	aws w; memset(&w, 0, sizeof w);
	aln r = band < 0 ? nw_score(&w, tseq, q, k, t_s, t_e, q_s, q_e, rw, tlen_total, NULL)
	                 : nw_band_score(&w, tseq, q, k, t_s, t_e, q_s, q_e, band, rw, tlen_total, NULL);
	out[0] = r.score; out[1] = r.len; out[2] = r.pos; out[3] = r.match; out[4] = r.tGaps; out[5] = r.qGaps;
	for(int i = 0; i < 2; ++i) { free(w.D[i]); free(w.P[i]); }
	free(w.E);
}

/* One read of stage 3c (assemble_KMA, assembly.c:1917-1965): the read as ConClave filed it under template t (bytes 0-4,
 * already reverse-complemented if its hit was on the reverse strand, conclave.c:131-146), aligned with KMA(); +Wl for
 * an alignment that starts at the first / ends at the last template base; kept if minlen <= aln_len, mrcheck, 0 < score
 * and scoreT <= score / aln_len. Outputs: stats[10] = {score, start, end, aln_len, clip_start, clip_end, match, tGaps,
 * qGaps, mapQ};
 * cols (capacity cap) receives the columns as '=' 'X' 'I' (gap in template) 'D' (gap in read), the classes makeCigar
 * uses (sam.c:57-78). Returns the number of columns if the read is kept, 0 if it is dropped, -needed if cap is too small. */
static int trace_result(aln S, const orc_rewards *rw, const orc_align_params *ap, int q_len, int t_len, unsigned mapQ, const trace *alp,
                        int *stats, char *cols, int cap);

int orc_align_trace(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap, const uint8_t *read, int q_len, int t,
                    int *stats, char *cols, int cap) {
	static trace al, fr;              /* oracle = single-threaded test code */
	const orc_db *db = a->db;
	const int k = db->kmersize, t_len = db->tlen[t];
	const uint64_t *tseq = db->tseq + db->tseq_off[t];
	memset(stats, 0, 10 * sizeof(int));
	unsigned mapQ = 0;
	if(!a->ix[t].o) tindex_build(&a->ix[t], tseq, t_len, k);
	fr.pos = 0;
	aln S = kma_trace(&a->w, &a->ix[t], tseq, t_len, k, read, q_len, 0, q_len, ap->mq, rw, &al, &fr, &mapQ, 0);
	return trace_result(S, rw, ap, q_len, t_len, mapQ, &al, stats, cols, cap);
}

/* the read filter of assemble_KMA (assembly.c:1931-1961) on KMA()'s result + the columns as classes */
static int trace_result(aln S, const orc_rewards *rw, const orc_align_params *ap, int q_len, int t_len, unsigned mapQ, const trace *alp,
                        int *stats, char *cols, int cap) {
	const trace al = *alp;
	const int aln_len = S.len, start = S.pos;
	int end = start + aln_len - S.tGaps;
	if(t_len < end) end -= t_len;
	int read_score = S.score;
	if(start == 0) read_score += rw->Wl;
	if(end == t_len) read_score += rw->Wl;
	double score;
	if(ap->minlen <= aln_len && ((ap->mrc * q_len <= S.len - S.qGaps) || (ap->mrc * t_len <= S.len - S.tGaps))) score = 1.0 * read_score / aln_len;
	else { read_score = 0; score = 0; }
	if(!(0 < read_score && ap->scoreT <= score)) return 0;
	stats[0] = read_score; stats[1] = start; stats[2] = (t_len < end) ? end - t_len : end; stats[3] = aln_len;
	stats[4] = al.start; stats[5] = al.end; stats[6] = S.match; stats[7] = S.tGaps; stats[8] = S.qGaps; stats[9] = (int) mapQ;
	if(aln_len > cap) return -aln_len;
	for(int i = 0; i < aln_len; ++i) cols[i] = al.s[i] == '|' ? '=' : al.t[i] == 5 ? 'I' : al.q[i] == 5 ? 'D' : 'X';
	return aln_len;
}

/* ---- `-Mt1`: anker_rc, align.c:780-991 -- both strands of a raw read are seeded against the one template, the strand
 * with the larger MEM coverage wins (forward on equality) and its MEMs are left in the points for KMA(). Byte-wise like
 * the reference: qseq is reverse-complemented in place (strrc, stdnuc.c:450-466) and stays that way when the reverse
 * strand wins. preseed (align.c:750-768, unless -ex_mode): when none of the read's every-k-th k-mers is in the index
 * the forward strand is not seeded at all. Returns bestScore; *is_rc = 1 when qseq is left reverse-complemented. */
static void strrc_bytes(uint8_t *q, int n) {
	static const uint8_t comp[6] = {3, 2, 1, 0, 4, 5};
	for(int i = 0, j = n - 1; i < (n >> 1); ++i, --j) { const uint8_t c = comp[q[i]]; q[i] = comp[q[j]]; q[j] = c; }
	if(n & 1) q[n >> 1] = comp[q[n >> 1]];
}

static int anker_rc_bytes(aws *w, const tindex *ix, const uint64_t *tseq, int t_len, int k, uint8_t *qseq, int q_len,
                          int one2one, int exhaustive, int *is_rc) {
	const uint64_t mask = (k < 32) ? ((1ull << (2 * k)) - 1) : ~0ull;
	int bestScore = 0, score = 0, score_r = 0, mem_count = 0, tot = 0, plen = 0, q_start = 0, q_end = q_len;
	*is_rc = 0;
	for(int rc = 0; rc < 2; ++rc) {
		int i;
		if(rc) {
			strrc_bytes(qseq, q_len);
			score = score_r; plen = mem_count;
			i = q_len - q_start; q_start = q_len - q_end; q_end = i; i = q_start;
		} else if(exhaustive) i = 0;
		else {
			/* bytes past the read end are whatever the reference's buffer held -- taken as 0 here */
			int hit = 0;
			for(i = 0; i < q_end - q_start && !hit; i += k) {
				uint64_t key = 0;
				for(int x = 0; x < k; ++x) key = (x ? (key << 2) : 0) | (uint64_t) ((i + x < q_len) ? qseq[i + x] : 0);
				if(tindex_has(ix, key)) hit = 1;
			}
			if(hit) i = 0;
		}
		score_r = 0; mem_count = 0;
		while(i < q_end) {
			int end = -1;
			for(int x = i; x < q_len; ++x) if(qseq[x] == 4) { end = x; break; }
			if(end == -1) end = q_end;
			uint64_t key = 0;
#define RESTART_KEY() do { if(i < end - k) { key = 0; for(int x = 0; x < k - 1; ++x) key = (key << 2) | qseq[i + x]; i += k - 1; } else i = end + 1; } while(0)
			RESTART_KEY();
			while(i < end) {
				key = ((key << 2) | qseq[i]) & mask;
				int cnt = 0;
				const int first = key ? tindex_find(ix, key, &cnt) : 0;
				if(cnt == 0) { ++i; continue; }
				i -= k - 1;
				if(cnt == 1) {
					aws_points(w, tot + 2);
					int value = ix->o[first].pos, prev = value - 2, j;
					for(j = i - 1; 0 <= j && 0 <= prev && qseq[j] == tnuc(tseq, prev); --j) { --prev; ++score_r; }
					w->qS[tot] = j + 1; w->tS[tot] = prev + 2;
					value += k - 1; i += k; score_r += k;
					while(i < end && value < t_len && qseq[i] == tnuc(tseq, value)) { ++i; ++value; ++score_r; }
					w->qE[tot] = i; w->tE[tot] = value + 1;
					w->w[tot] = w->tE[tot] - w->tS[tot];
					++mem_count; ++tot;
				} else {
					score_r += k;
					int bias = i;
					for(int c = 0; c < cnt; ++c) {
						aws_points(w, tot + 2);
						int value = ix->o[first + c].pos, prev = value - 2, j, kk = i;
						for(j = kk - 1; 0 <= j && 0 <= prev && qseq[j] == tnuc(tseq, prev); --j) --prev;
						w->qS[tot] = j + 1; w->tS[tot] = prev + 2;
						value += k - 1; kk += k;
						while(kk < end && value < t_len && qseq[kk] == tnuc(tseq, value)) { ++kk; ++value; }
						w->qE[tot] = kk; w->tE[tot] = value + 1;
						w->w[tot] = w->qE[tot] - w->qS[tot];
						++mem_count; ++tot;
						if(bias < kk) bias = kk;
					}
					score_r += bias - i;
					i = bias + 1;
				}
				RESTART_KEY();
			}
#undef RESTART_KEY
			i = end + 1;
		}
		if(bestScore < score_r) bestScore = score_r;
	}
	if(one2one && bestScore < k && bestScore * k < (q_len - k - bestScore)) { w->plen = 0; *is_rc = 1; return 0; }
	if(bestScore == score) { strrc_bytes(qseq, q_len); w->plen = plen; return bestScore; }
	*is_rc = 1;
	if(plen) {
		for(int x = 0; x < mem_count; ++x) {
			w->tS[x] = w->tS[plen + x]; w->tE[x] = w->tE[plen + x]; w->qS[x] = w->qS[plen + x];
			w->qE[x] = w->qE[plen + x]; w->w[x] = w->w[plen + x];
		}
	}
	w->plen = mem_count;
	return bestScore;
}

/* One raw read of the `-Mt1 t` run (runKMA_Mt1, mt1.c:86-500 -> assemble_KMA, assembly.c:1917-1965 with read_score == 0):
 * anker_rc picks the strand and leaves the MEMs, KMA() chains and joins them, then the read filter. `read` (codes 0-4,
 * as stage 1 trimmed it) is overwritten with the orientation that was aligned; *is_rc says which. Other outputs as
 * orc_align_trace. */
int orc_align_trace_mt1(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap, uint8_t *read, int q_len, int t,
                        int one2one, int exhaustive, int *stats, char *cols, int cap, int *is_rc) {
	static trace al, fr;
	const orc_db *db = a->db;
	const int k = db->kmersize, t_len = db->tlen[t];
	const uint64_t *tseq = db->tseq + db->tseq_off[t];
	memset(stats, 0, 10 * sizeof(int));
	if(!a->ix[t].o) tindex_build(&a->ix[t], tseq, t_len, k);
	if(!anker_rc_bytes(&a->w, &a->ix[t], tseq, t_len, k, read, q_len, one2one, exhaustive, is_rc)) return 0;
	unsigned mapQ = 0;
	fr.pos = 0;
	aln S = kma_trace(&a->w, &a->ix[t], tseq, t_len, k, read, q_len, 0, q_len, ap->mq, rw, &al, &fr, &mapQ, 1);
	return trace_result(S, rw, ap, q_len, t_len, mapQ, &al, stats, cols, cap);
}

int64_t orc_align_se_batch(const orc_db *db, const orc_rewards *rw, const orc_align_params *ap,
                           int64_t n_reads, const uint64_t *seq, const int64_t *seq_off,
                           const int32_t *len, const int32_t *N, const int64_t *N_off,
                           const int32_t *rc_flag, const int32_t *flag, const int64_t *T_off, const int32_t *T,
                           int32_t *n_hits, int32_t *best_score, int32_t *out_flag,
                           int32_t *ht, int32_t *hs, int32_t *he, int32_t *hscore,
                           uint64_t *alignment_scores, uint64_t *uniq_alignment_scores) {
	/* hits of read r are written at [T_off[r], T_off[r] + n_hits[r]) */
	orc_aligner *a = orc_aligner_new(db);
	int64_t mapped = 0;
	for(int64_t r = 0; r < n_reads; ++r) {
		const int words = (len[r] + 31) >> 5;
		uint64_t *s = malloc(((size_t) words + 2) * 8);
		memcpy(s, seq + seq_off[r], (size_t) words * 8); s[words] = 0; s[words + 1] = 0;
		const int64_t o = T_off[r];
		int nh = 0, bs = 0, of = 0;
		orc_align_se(a, rw, ap, s, len[r], N + N_off[r], (int) (N_off[r + 1] - N_off[r]), rc_flag[r], flag[r],
		             T + o, (int) (T_off[r + 1] - o), &nh, &bs, &of, ht + o, hs + o, he + o, hscore + o,
		             alignment_scores, uniq_alignment_scores);
		n_hits[r] = nh; best_score[r] = bs; out_flag[r] = of;
		if(nh > 0) ++mapped;
		free(s);
	}
	orc_aligner_free(a);
	return mapped;
}

/* ---- paired end: alnFragsPenaltyPE (alnfrags.c:1596-1972), update_Scores_pe (updatescores.c:390-488),
 * update_Scores_se (:300-388). seqA/seqB are the two S2 records AS WRITTEN to the stream (first record with
 * the empty list, second with the candidates T). No strand-tie (arc) handling: save_kmers_penaltyPair never
 * writes a negative score on the first record of a proper pair. */
static void orient(orc_aligner *a, const uint64_t *seq, int len, const int *N, int nN, int rc,
                   uint8_t *bytes, uint64_t *comp, int *Nout) {
	int *tmp = malloc(sizeof(int) * (size_t) (nN + 3));
	tmp[0] = nN; memcpy(tmp + 1, N, sizeof(int) * (size_t) nN);
	const int words = (len + 31) >> 5;
	if(rc) { orc_rc(seq, len, tmp, comp, Nout); }
	else { memcpy(comp, seq, sizeof(uint64_t) * (size_t) words); memcpy(Nout, tmp, sizeof(int) * (size_t) (nN + 1)); }
	comp[words] = 0; comp[words + 1] = 0;
	unpack_bytes(comp, len, Nout, bytes);
	Nout[0] += 1; Nout[Nout[0]] = len;
	free(tmp);
	(void) a;
}

int orc_align_pe(orc_aligner *a, const orc_rewards *rw, const orc_align_params *ap,
                 const uint64_t *seqA, int lenA, const int *NA, int nNA, int flagA,
                 const uint64_t *seqB, int lenB, const int *NB, int nNB, int flagB,
                 const int *T, int nT, orc_pe_out *out,
                 uint64_t *alignment_scores, uint64_t *uniq_alignment_scores) {
	const orc_db *db = a->db;
	const int k = db->kmersize, Wl = -rw->Wl, PE = rw->PE;
	{ int *t = out->tmpl, *sc = out->score, *st = out->start, *en = out->end; memset(out, 0, sizeof *out); out->tmpl = t; out->score = sc; out->start = st; out->end = en; }
	out->flagA = flagA; out->flagB = flagB;
	uint8_t *qa = malloc((size_t) lenA + 2), *qb = malloc((size_t) lenB + 2);
	uint64_t *ca = calloc((size_t) ((lenA + 31) >> 5) + 3, 8), *cb = calloc((size_t) ((lenB + 31) >> 5) + 3, 8);
	int *Na = malloc(sizeof(int) * (size_t) (nNA + 4)), *Nb = malloc(sizeof(int) * (size_t) (nNB + 4));
	int *bt = calloc((size_t) nT + 2, sizeof(int)), *btr = calloc((size_t) nT + 2, sizeof(int));
	int *bs = calloc((size_t) nT + 2, sizeof(int)), *be = calloc((size_t) nT + 2, sizeof(int));
	int rcstate = 0, best = 0, best_r = 0, comp = 0;
	orient(a, seqA, lenA, NA, nNA, 0, qa, ca, Na);
	orient(a, seqB, lenB, NB, nNB, 0, qb, cb, Nb);
	for(int i = 0; i < nT; ++i) {
		const int tmpl = T[i], at = abs(tmpl);
		if(tmpl < 0 && !rcstate) {   /* alnfrags.c:1633-1647: both mates are flipped once and stay flipped */
			orient(a, seqA, lenA, NA, nNA, 1, qa, ca, Na);
			orient(a, seqB, lenB, NB, nNB, 1, qb, cb, Nb);
			rcstate = 1;
		}
		const int t_len = db->tlen[at];
		const uint64_t *tseq = db->tseq + db->tseq_off[at];
		if(!a->ix[at].o) tindex_build(&a->ix[at], tseq, t_len, k);
		int start = 0, end = 0;
		for(int m = 0; m < 2; ++m) {
			const int qlen = m ? lenB : lenA;
			aln st = kma_score(&a->w, &a->ix[at], tseq, t_len, k, m ? qb : qa, qlen, 0, qlen, m ? cb : ca, m ? Nb : Na, ap->mq, rw, 0);
			int rs = st.score; double score = 0;
			if(ap->minlen <= st.len && 0 < rs && ((ap->mrc * qlen <= st.len - st.qGaps) || (ap->mrc * t_len <= st.len - st.tGaps))) {
				start = st.pos; end = st.pos + st.len - st.tGaps;
				if(start == 0) rs += Wl;
				if(end == t_len) rs += Wl;
				score = 1.0 * rs / st.len;
			} else rs = 0;
			const int ok = rs > k && score >= ap->scoreT;
			if(m == 0) {
				if(ok) { bt[i] = rs; bs[i] = start; be[i] = end; if(best < rs) best = rs; }
				else { bt[i] = 0; bs[i] = -1; be[i] = -1; }
			} else {
				if(ok) {
					btr[i] = rs;
					if(bt[i]) { if(start < bs[i]) bs[i] = start; else be[i] = end; }
					else { bs[i] = start; be[i] = end; }
					if(best_r < rs) best_r = rs;
				} else btr[i] = 0;
				/* alnfrags.c:1771: the raw second-mate score (0 if its own test failed) joins the first mate's */
				const int both = rs + bt[i];
				if(comp < both) comp = both;
			}
		}
	}
	int ret = 3;
	if(best || best_r) {
		const double need = 1.0 * (best + best_r);   /* minFrac == 1 */
		if(comp && need <= comp + PE) {
			/* proper pair */
			const int bestScore = comp + PE;
			int h = 0;
			for(int i = 0; i < nT; ++i) if(bt[i] && btr[i]) { out->score[h] = bt[i] + btr[i] + PE; out->tmpl[h] = T[i]; out->start[h] = bs[i]; out->end[h] = be[i]; ++h; }
			int swapped = 0;
			if(h && out->tmpl[0] < 0) { for(int i = 0; i < h; ++i) out->tmpl[i] = -out->tmpl[i]; swapped = 1; }
			else if(rcstate) { out->flagA ^= 48; out->flagB ^= 48; }
			/* update_Scores_pe, minFrac == 1 */
			int c = 0;
			for(int i = 0; i < h; ++i) if(out->score[i] == bestScore) {
				out->tmpl[c] = out->tmpl[i]; out->start[c] = out->start[i]; out->end[c] = out->end[i]; ++c;
				if(alignment_scores) alignment_scores[abs(out->tmpl[c - 1])] += (uint64_t) bestScore;
			}
			if(c == 1 && uniq_alignment_scores) uniq_alignment_scores[abs(out->tmpl[0])] += (uint64_t) bestScore;
			out->kind = 1; out->swapped = swapped; out->n_hits = c; out->best = bestScore; out->best_r = bestScore;
			/* orientation of the two sequences as stored in the frag record */
			out->rcA = swapped ? rcstate : 0; out->rcB = swapped ? rcstate : 0;
			ret = 0;
		} else if(best && best_r) {
			/* unmated pair (alnfrags.c:1820-1891), restated literally on 1-based arrays incl. the
			 * `+ end` pointer shift; hits of the first record go to [0, n_hits), of the second after them */
			int *mt = malloc(sizeof(int) * (size_t) (nT + 2));
			int *b1 = malloc(sizeof(int) * (size_t) (nT + 2)), *b2 = malloc(sizeof(int) * (size_t) (nT + 2));
			int *s1 = malloc(sizeof(int) * (size_t) (nT + 2)), *e1 = malloc(sizeof(int) * (size_t) (nT + 2));
			for(int i = 0; i < nT; ++i) { mt[i + 1] = T[i]; b1[i + 1] = bt[i]; b2[i + 1] = btr[i]; s1[i + 1] = bs[i]; e1[i + 1] = be[i]; }
			mt[0] = nT; b1[0] = b2[0] = s1[0] = e1[0] = 0;
			int h = 0, hr = 0, ti = 1, endp = nT, x;
			const double sc1 = best, sc2 = best_r;
			while(ti <= endp) {
				if(sc1 <= b1[ti]) { mt[h] = mt[ti]; b1[h] = b1[ti]; s1[h] = s1[ti]; e1[h] = e1[ti]; ++h; ++ti; }
				else if(sc2 <= b2[ti]) {
					x = mt[ti]; mt[ti] = mt[endp]; mt[endp] = x; x = b2[ti]; b2[ti] = b2[endp]; b2[endp] = x;
					x = s1[ti]; s1[ti] = s1[endp]; s1[endp] = x; x = e1[ti]; e1[ti] = e1[endp]; e1[endp] = x;
					++hr; --endp;
				} else ++ti;
			}
			int fA = flagA, fB = flagB;
			if(b1[0] < 0) { for(int i = 0; i < h; ++i) b1[i] = -b1[i]; }
			else if(rcstate) { fA ^= 16; fB ^= 32; }
			if(b2[endp] < 0) { for(int i = 0; i < hr; ++i) b2[endp + i] = -b2[endp + i]; }
			else if(rcstate) { fA ^= 32; fB ^= 16; }
			if(fA & 2) { fA ^= 2; fB ^= 2; }
			int c = 0;
			for(int i = 0; i < h; ++i) if(b1[i] == best) {
				out->tmpl[c] = mt[i]; out->start[c] = s1[i]; out->end[c] = e1[i]; out->score[c] = best; ++c;
				if(alignment_scores) alignment_scores[abs(mt[i])] += (uint64_t) best;
			}
			if(c == 1 && uniq_alignment_scores) uniq_alignment_scores[abs(out->tmpl[0])] += (uint64_t) best;
			int c2 = 0;
			for(int i = 0; i < hr; ++i) if(b2[endp + i] == best_r) {
				out->tmpl[c + c2] = mt[endp + i]; out->start[c + c2] = s1[endp + i]; out->end[c + c2] = e1[endp + i]; out->score[c + c2] = best_r; ++c2;
				if(alignment_scores) alignment_scores[abs(mt[endp + i])] += (uint64_t) best_r;
			}
			if(c2 == 1 && uniq_alignment_scores) uniq_alignment_scores[abs(out->tmpl[c])] += (uint64_t) best_r;
			out->kind = 2; out->n_hits = c; out->n_hits_r = c2; out->best = best; out->best_r = best_r; out->flagA = fA; out->flagB = fB;
			free(mt); free(b1); free(b2); free(s1); free(e1);
			ret = 0;
		} else {
			/* only one record aligned (alnfrags.c:1892-1966) */
			const int first = best != 0;
			const int *sc = first ? bt : btr;
			const int bscore = first ? best : best_r;
			int h = 0;
			for(int i = 0; i < nT; ++i) if(sc[i]) { out->score[h] = sc[i]; out->tmpl[h] = T[i]; out->start[h] = bs[i]; out->end[h] = be[i]; ++h; }
			int fA = flagA, fB = flagB;
			if(first) {
				if(h && out->tmpl[0] < 0) { for(int i = 0; i < h; ++i) out->tmpl[i] = -out->tmpl[i]; }
				else if(rcstate) { fA ^= 16; fB ^= 32; }
				fA |= 8; fB ^= 4;
				if(fA & 2) { fA ^= 2; fB ^= 2; }
			} else {
				/* the reference tests the sign of a SCORE here (never negative): templates keep their sign */
				if(rcstate) { fA ^= 32; fB ^= 16; }
				fB |= 8; fA ^= 4;
				if(fB & 2) { fA ^= 2; fB ^= 2; }
			}
			int c = 0;
			for(int i = 0; i < h; ++i) if(out->score[i] == bscore) {
				out->tmpl[c] = out->tmpl[i]; out->start[c] = out->start[i]; out->end[c] = out->end[i]; ++c;
				if(alignment_scores) alignment_scores[abs(out->tmpl[c - 1])] += (uint64_t) bscore;
			}
			if(c == 1 && uniq_alignment_scores) uniq_alignment_scores[abs(out->tmpl[0])] += (uint64_t) bscore;
			out->kind = first ? 3 : 4; out->n_hits = c; out->best = best; out->best_r = best_r; out->flagA = fA; out->flagB = fB;
			ret = first ? 2 : 1;
		}
	}
	free(qa); free(qb); free(ca); free(cb); free(Na); free(Nb); free(bt); free(btr); free(bs); free(be);
	return ret;
}
