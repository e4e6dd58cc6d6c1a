/* oracle/chain.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of KMA's default template finder, save_kmers_chain (savekmers.c:5127-5945) with the default function
 * pointers (kmeranker.c:25-30: getBestChainTemplates, ankerScore, testExtensionScore, proxiTestBestScore, getBestAnkerScore,
 * getTieAnkerScore), chooseChain with proxi 1.0 (kmeranker.c:512-595), mrchain with mrc 0 (kmeranker.c:57-81: always 1),
 * pruneAnkers (:372-398) and the segment tree of seqmenttree.c:25-233. One read in, zero or more S2 records out: every accepted
 * chain is its own record with its query bounds (insertKmerBound, qseqs.c:41-56).
 *
 * Anchors = maximal runs of k-mer starts whose value list is the same (pointer identity in the reference = equal list offset
 * here), per strand, both in FORWARD read coordinates. Parity: pinned by tests/test_oracle_golden.py against the reference's
 * `-s2` tap without `-1t1` (tests/golden/se, long). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "kma_oracle.h"

static inline uint64_t kmer_at_pub(const uint64_t *seq, int pos, int k) {
	/* stdnuc.h:27-30 (getKmer_macro); may touch seq[word + 1] */
	const int ip = (pos & 31) << 1, w = pos >> 5, sh = 64 - (k << 1);
	if(ip <= sh) return (seq[w] << ip) >> sh;
	return ((seq[w] << ip) | (seq[w + 1] >> (64 - ip))) >> sh;
}

typedef struct {
	int score, weight, score_len, len_len;
	unsigned start, end;
	int64_t values;      /* offset of the value list, -1 none */
	int descend;         /* index of the next anchor of the strand, -1 none */
} anker;

/* seqmenttree.c: nodes in an array, children by index (-1 none). The reference keeps pointers and grows by copying; the
 * shapes and the covered counts are what matters. */
typedef struct { unsigned start, end, covered; int b0, b1; } seg;
typedef struct { seg *v; int n, cap; } segtree;

static unsigned seg_add(segtree *t, int root, int node) {
	seg *R = &t->v[root], *Nn = &t->v[node];
	if(R->b0 >= 0) {
		if(Nn->start < R->start && R->end < Nn->end) {
			R->start = Nn->start; R->end = Nn->end; R->covered = Nn->covered; Nn->covered = 0; R->b0 = -1;
			return R->covered;
		} else if(R->end < Nn->end) R->end = Nn->end;
		else if(Nn->start < R->start) R->start = Nn->start;
		unsigned pos = t->v[R->b1].start;
		if(Nn->end < pos) R->covered = t->v[R->b1].covered + seg_add(t, R->b0, node);
		else if(pos <= Nn->start) R->covered = t->v[R->b0].covered + seg_add(t, R->b1, node);
		else {
			pos = Nn->start;
			Nn->start = t->v[R->b0].end + 1;
			Nn->covered = Nn->end - Nn->start;
			unsigned covered = seg_add(t, R->b1, node);
			Nn->start = pos;
			Nn->end = t->v[R->b0].end;
			Nn->covered = Nn->end - Nn->start;
			R->covered = covered + seg_add(t, R->b0, node);
		}
	} else if(Nn->end < R->start || R->end < Nn->start) {
		const int bud = node + 1;
		t->v[bud].start = R->start; t->v[bud].end = R->end; t->v[bud].covered = R->covered; t->v[bud].b0 = -1;
		if(Nn->end < R->start) { R->start = Nn->start; R->b0 = node; R->b1 = bud; }
		else { R->end = Nn->end; R->b0 = bud; R->b1 = node; }
		R->covered += Nn->covered;
	} else {
		if(Nn->start < R->start) R->start = Nn->start;
		if(R->end < Nn->end) R->end = Nn->end;
		Nn->covered = 0;
		R->covered = R->end - R->start;
	}
	return R->covered;
}

static void seg_grow(segtree *t, unsigned start, unsigned end) {
	if(t->cap <= t->n + 2) { t->cap = t->cap ? 2 * t->cap : 64; t->v = realloc(t->v, (size_t) t->cap * sizeof(seg)); }
	if(t->n == 0) {
		t->n = 1;
		t->v[0].start = start; t->v[0].end = end; t->v[0].covered = end - start; t->v[0].b0 = t->v[0].b1 = -1;
		return;
	}
	const int node = t->n;
	t->v[node].start = start; t->v[node].end = end; t->v[node].covered = end - start; t->v[node].b0 = -1;
	t->v[0].covered = seg_add(t, 0, node);
	if(t->v[node].covered) t->n += 2;
}

static unsigned seg_que(const segtree *t, int i, unsigned start, unsigned end) {
	const seg *s = &t->v[i];
	if(end < s->start || s->end < start) return 0;
	if(start <= s->start && s->end <= end) return s->covered;
	if(s->b0 >= 0) return seg_que(t, s->b0, start, end) + seg_que(t, s->b1, start, end);
	if(s->start <= start && end <= s->end) return end - start;
	if(s->start <= start && start < s->end) return s->end - start;
	if(s->start < end && end <= s->end) return end - s->start;
	return 0;
}

typedef struct {
	const orc_db *db;
	const orc_rewards *rw;
	const int *tlen;
	int q_len, k;
	int *Score, *extend;    /* DB_size */
	char *include;          /* DB_size; include[0] unused here (the reference keeps its u16 flag there) */
} chain_ctx;

static int list_n(const orc_db *db, int64_t v) { return db->values16 ? db->values16[v] : (int) db->values32[v]; }
static int list_at(const orc_db *db, int64_t v, int i) { return db->values16 ? db->values16[v + i] : (int) db->values32[v + i]; }

/* the bridge between two anchors of one template, chain direction given by the caller (gaps between them) */
static int bridge(const orc_rewards *rw, int k, int mlen, int weight, int gaps) {
	const int M = rw->M, MM = rw->MM, U = rw->U, W1 = rw->W1;
	if(gaps == -k) return weight - (k - 1) * M;
	if(gaps == 0) return weight + MM;
	if(0 < gaps) {
		int MMs, Ms;
		if(gaps <= 2) { MMs = gaps; Ms = 0; }
		else {
			MMs = gaps / k + (gaps % k ? 1 : 0);
			if(MMs < 2) MMs = 2;
			Ms = gaps - MMs < k ? gaps - MMs : k;
			if(MMs < Ms) Ms = MMs;
		}
		if(W1 + (gaps - 1) * U <= MMs * MM + Ms * M) return weight + Ms * M + MMs * MM;
		return weight + W1 + (gaps - 1) * U;
	}
	if(mlen != k) return weight + gaps * M + MM;
	return weight + gaps * M - (gaps + 1) * U + W1;
}

/* getBestChainTemplates, kmeranker.c:83-234: the templates of the chain that ends in V[src]; walks DOWN the anchor array,
 * silences the anchors it passes (score = 0), returns the index of the anchor the chain starts at, or -1 */
static int chain_templates(chain_ctx *c, anker *V, int src, int *bests) {
	const orc_db *db = c->db;
	const orc_rewards *rw = c->rw;
	const int k = c->k;
	int nextAnker = 0;
	if(src < 0) return -1;
	{
		const int n = list_n(db, V[src].values);
		bests[0] = n;
		for(int i = n; i >= 1; --i) {
			const int t = list_at(db, V[src].values, i);
			bests[i] = t;
			if(++c->include[t] == 1) nextAnker = 1;
		}
	}
	const int bestScore = V[src].score;
	int prev = src;
	for(int node = src; nextAnker && node >= 0; --node) {
		const int n = list_n(db, V[node].values);
		const int start = (int) V[node].start, end = (int) V[node].end;
		for(int i = n; i >= 1; --i) {
			const int t = list_at(db, V[node].values, i);
			if(!c->include[t]) continue;
			int score = c->Score[t];
			const int pos = c->extend[t];
			const int gaps = pos - end;
			if(pos == 0) score = V[node].weight;
			else {
				score += bridge(rw, k, (int) db->mlen, V[node].weight, gaps);
				V[node].score = 0;
			}
			if(bestScore <= score) {
				int tmp;
				if(V[node].start) {
					tmp = rw->W1 + ((int) V[node].start - 1) * rw->U;
					tmp = score + (rw->Wl < tmp ? tmp : rw->Wl);
				} else tmp = score;
				if(tmp == bestScore) { score = bestScore; nextAnker = 0; prev = node; }
			}
			c->extend[t] = start;
			c->Score[t] = score;
		}
	}
	int j = 0;
	for(int i = 1; i <= bests[0]; ++i) {
		const int t = bests[i];
		if(c->include[t] == 1 && bestScore <= c->Score[t]) bests[++j] = t;
		c->Score[t] = 0; c->include[t] = 0; c->extend[t] = 0;
	}
	bests[0] = j;
	return j ? prev : -1;
}

/* pruneAnkers, kmeranker.c:372-398: head of the list of anchors with score >= k, -1 none */
static int prune(anker *V, int head, int k) {
	while(head >= 0 && V[head].score < k) head = V[head].descend;
	if(head < 0) return -1;
	int prev = head;
	for(int node = V[head].descend; node >= 0; node = V[node].descend) {
		if(k <= V[node].score) { V[prev].descend = node; prev = node; }
	}
	V[prev].descend = -1;
	return head;
}

/* getBestAnkerScore, kmeranker.c:400-431 */
static int best_anker(anker *V, int *head, unsigned *ties) {
	*ties = 0;
	int prev = *head;
	while(prev >= 0 && V[prev].score == 0) prev = V[prev].descend;
	*head = prev;
	if(prev < 0) return -1;
	int best = prev;
	for(int node = V[prev].descend; node >= 0; node = V[node].descend) {
		if(V[node].score) {
			if(V[best].score < V[node].score) { best = node; *ties = 0; }
			else if(V[best].score == V[node].score) { best = node; ++*ties; }
			V[prev].descend = node;
			prev = node;
		}
	}
	V[prev].descend = -1;
	return best;
}

/* getTieAnkerScore, kmeranker.c:477-492 */
static int tie_anker(const anker *V, int stop, int src, int best) {
	if(src < 0 || (int) V[src].start <= stop) return -1;
	while(stop < (int) V[--src].start) if(V[src].score == V[best].score) return src;
	return -1;
}

/* chooseChain with proxi == 1.0, kmeranker.c:512-595 */
static int choose_chain(const anker *b, const anker *r, int cStart, int cStart_r, double coverT, int *Start, int *Len) {
	int rc = r->score < b->score ? 1 : b->score < r->score ? 2 : 3, start, end;
	if(rc == 1) { start = cStart; end = (int) b->end; }
	else if(rc == 2) { start = cStart_r; end = (int) r->end; }
	else if((int) b->end < cStart_r) { start = cStart; end = (int) b->end; rc = 1; }
	else if((int) r->end < cStart) { start = cStart_r; end = (int) r->end; rc = 2; }
	else if(cStart <= cStart_r && r->end <= b->end) { start = cStart; end = (int) b->end; }
	else if(cStart_r <= cStart && b->end <= r->end) { start = cStart_r; end = (int) r->end; }
	else if(r->end < b->end) {
		start = (int) b->end - cStart;
		end = (int) r->end - cStart_r;
		end = start < end ? start : end;
		start = cStart_r;
		if(coverT * end <= (int) r->end - cStart) end = (int) b->end;
		else { end = (int) r->end; rc = 2; }
	} else {
		start = (int) b->end - cStart;
		end = (int) r->end - cStart_r;
		end = start < end ? start : end;
		start = cStart;
		if(coverT * end <= (int) b->end - cStart_r) end = (int) r->end;
		else { end = (int) b->end; rc = 1; }
	}
	*Start = start; *Len = end - start;
	return rc;
}

/* anchors of one strand in forward coordinates (savekmers.c:5208-5330 forward, :5333-5452 reverse: the k-mer at forward
 * position j is looked up reverse-complemented). Returns the number of anchors. */
static int build_ankers(const orc_db *db, const orc_rewards *rw, int exhaustive, int is_rc, const uint64_t *seq, const uint64_t *rseq,
                        int seqlen, const int *N, int nN, const int *rN, anker *V) {
	const int k = (int) db->kmersize, M = rw->M, MM = rw->MM, U = rw->U, W1 = rw->W1;
	int hits = 0;
	V[0].start = 0; V[0].end = 0; V[0].values = -1; V[0].descend = -1;
	/* prefilter: every k-th k-mer of every N-free segment of the strand that is scanned */
	int HIT = exhaustive;
	{
		const uint64_t *s = is_rc ? rseq : seq;
		const int *Ns = is_rc ? rN : N;
		int j = 0;
		for(int i = 1; i <= nN + 1 && !HIT; ++i) {
			const int segend = i <= nN ? Ns[i] : seqlen;
			for(; j < segend - k + 1 && !HIT; j += k) if(orc_hash_get(db, kmer_at_pub(s, j, k)) >= 0) HIT = 1;
			j = segend + 1;
		}
	}
	if(!HIT) return 0;
	int v = 0, Ms = 0, MMs = 0, Us = 0, W1s = 0, gaps = 0, j = 0;
	int64_t last = -1;
	const int seqend = seqlen - k + 1;
	/* the reverse strand's k-mer for forward position j starts at rcpos in the reverse complement: seqlen - k - j up to the
	 * first N; behind an N the reference restarts it at seqlen - j (savekmers.c:5447-5449), k further on -- kept as it is */
	int rcpos = seqlen - k;
	for(int i = 1; i <= nN + 1 && j < seqend; ++i) {
		const int segend = i <= nN ? N[i] : seqlen;
		for(; j < segend - k + 1; ++j, --rcpos) {
			const uint64_t km = is_rc ? (rcpos >= 0 ? kmer_at_pub(rseq, rcpos, k) : 0) : kmer_at_pub(seq, j, k);
			const int64_t values = orc_hash_get(db, km);
			if(values >= 0) {
				int open = 0;
				if(values == last) {
					if(gaps == 0) ++Ms;
					else if(gaps == k) { Ms += k; ++MMs; }
					else open = 1;
				} else open = 1;
				if(open) {
					if(last >= 0) {
						V[v].weight = Ms * M + MMs * MM + Us * U + W1s * W1;
						V[v].end = (unsigned) (j - gaps + k);
						V[v].descend = v + 1;
						++v;
					}
					V[v].start = (unsigned) j; V[v].values = values; V[v].descend = -1;
					last = values;
					Ms = k; MMs = 0; Us = 0; W1s = 0;
					++hits;
				}
				gaps = 0;
			} else ++gaps;
		}
		gaps += segend + 1 - j;
		j = segend + 1;
		rcpos = seqlen - j;
	}
	if(last >= 0) {
		V[v].weight = Ms * M + MMs * MM + Us * U + W1s * W1;
		V[v].end = (unsigned) (seqlen - gaps);
	}
	return hits;
}

int orc_scan_chain(const orc_db *db, const orc_rewards *rw, int exhaustive, int minlen, double coverT, double mrs,
                   const uint64_t *seq, int seqlen, const int *N /* N[0] = count */, orc_chain_rec *out, int out_cap, int *T_pool, int T_cap) {
	const int k = (int) db->kmersize, D = (int) db->DB_size;
	if(seqlen < k) return 0;
	const int nN = N[0];
	const int complen = (seqlen + 31) >> 5;
	uint64_t *rseq = calloc((size_t) complen + 2, 8);
	int *rN = calloc((size_t) nN + 2, sizeof(int));
	orc_rc(seq, seqlen, N, rseq, rN);
	anker *VF = calloc((size_t) seqlen + 2, sizeof(anker)), *VR = calloc((size_t) seqlen + 2, sizeof(anker));
	int *bestT = calloc((size_t) 2 * D + 4, sizeof(int)), *bestT_r = calloc((size_t) 2 * D + 4, sizeof(int));
	chain_ctx c;
	c.db = db; c.rw = rw; c.tlen = db->tlen; c.q_len = seqlen; c.k = k;
	c.Score = calloc((size_t) D + 1, sizeof(int)); c.extend = calloc((size_t) D + 1, sizeof(int)); c.include = calloc((size_t) D + 1, 1);
	segtree tree = {0, 0, 0};
	int n_out = 0, T_used = 0;
	const int M = rw->M, MM = rw->MM, U = rw->U, W1 = rw->W1, Wl = rw->Wl;
	(void) M; (void) MM;

	const unsigned hitF = (unsigned) build_ankers(db, rw, exhaustive, 0, seq, rseq, seqlen, N, nN, rN, VF);
	const unsigned hitR = (unsigned) build_ankers(db, rw, exhaustive, 1, seq, rseq, seqlen, N, nN, rN, VR);
	if(getenv("ORC_CHAIN_DEBUG")) {
		for(unsigned x = 0; x < hitF; ++x) fprintf(stderr, "F[%u] %u-%u w %d list %lld\n", x, VF[x].start, VF[x].end, VF[x].weight, (long long) VF[x].values);
		for(unsigned x = 0; x < hitR; ++x) fprintf(stderr, "R[%u] %u-%u w %d list %lld\n", x, VR[x].start, VR[x].end, VR[x].weight, (long long) VR[x].values);
	}
	if(!hitF && !hitR) goto done;

	/* chains, left to right, per strand (savekmers.c:5466-5634) */
	anker *best = 0, *best_r = &VF[0];
	unsigned ties = 0;
	VF[0].score = 0;
	{
		int *bests = bestT;
		bestT[0] = 0; bestT_r[0] = 0;
		for(int strand = 0; strand < 2; ++strand) {
			anker *V = strand ? VR : VF;
			unsigned HIT = (strand ? hitR : hitF) + 1;
			if(strand) {
				V[0].score = 0; V[0].score_len = 0; V[0].len_len = 1;
				bests = bestT_r;
				best = best_r;
				best_r = &V[0];
			}
			bests[0] = 0;
			int vi = 0;
			while(--HIT) {
				anker *A = &V[vi];
				const int start = (int) A->start, end = (int) A->end;
				A->score = 0; A->score_len = 0; A->len_len = 1;
				const int n = list_n(db, A->values);
				for(int i = n; i >= 1; --i) {
					const int t = list_at(db, A->values, i);
					int score = c.Score[t];
					const int pos = c.extend[t];
					const int gaps = start - pos;
					if(!c.include[t]) {
						c.include[t] = 1;
						bests[++bests[0]] = t;
						if(start) {
							score = W1 + (start - 1) * U;
							score = A->weight + (Wl < score ? score : Wl);
						} else score = A->weight;
					} else {
						score += bridge(rw, k, (int) db->mlen, A->weight, gaps);
						if(score < 0) {
							int test = start ? W1 + (start - 1) * U : 0;
							if(test < Wl) test = Wl;
							if(score < test + A->weight) score = test + A->weight;
						}
					}
					if(A->score < score) A->score = score;
					int len_len = c.tlen[t];
					if(seqlen < len_len) len_len = seqlen;
					double score_len = score;
					if(A->len_len != len_len) { score_len /= len_len; score_len *= A->len_len; }
					if(A->score_len < score_len || (A->score_len == score_len && A->score_len < score)) { A->score_len = score; A->len_len = len_len; }
					c.Score[t] = score;
					c.extend[t] = end;
				}
				/* last best hit of the strand (the length-corrected twin of this bookkeeping only matters with -ca) */
				if(best_r->score < A->score) { best_r = A; ties = 0; }
				else if(best_r->score == A->score) {
					if(best_r->score_len < A->score_len) { best_r = A; ties = 0; }
					else { best_r = A; ++ties; }
				}
				++vi;
			}
			for(int i = 1; i <= bests[0]; ++i) { c.Score[bests[i]] = 0; c.extend[bests[i]] = 0; c.include[bests[i]] = 0; }
		}
	}
	if(best->score < k && best_r->score < k) goto done;

	const int VF_start = (int) VF[0].start, VR_start = (int) VR[0].start;
	int headF = prune(VF, 0, k), headR = prune(VR, 0, k);
	if(headF < 0) best->score = 0;
	if(headR < 0) best_r->score = 0;
	bestT[0] = 0; bestT_r[0] = 0;
	int bi = (int) (best - VF), bri = (int) (best_r - VR);      /* -1 later: the strand is exhausted */
	int cStart = -1, cStart_r = -1, start = 0, len = 0, rc;
	if(!best->score || !best_r->score) {
		if(best->score) {
			const int s = chain_templates(&c, VF, bi, bestT);
			if(s < 0) goto done;
			cStart = (int) VF[s].start; start = cStart; len = (int) VF[bi].end - start; rc = 1;
		} else {
			const int s = chain_templates(&c, VR, bri, bestT_r);
			if(s < 0) goto done;
			cStart_r = (int) VR[s].start; start = cStart_r; len = (int) VR[bri].end - start; rc = 2;
		}
	} else {
		int s = chain_templates(&c, VF, bi, bestT);
		if(s < 0) goto done;
		cStart = (int) VF[s].start;
		s = chain_templates(&c, VR, bri, bestT_r);
		if(s < 0) goto done;
		cStart_r = (int) VR[s].start;
		rc = choose_chain(&VF[bi], &VR[bri], cStart, cStart_r, coverT, &start, &len);
	}
	{
		const int score = VF[bi].score > VR[bri].score ? VF[bi].score : VR[bri].score;
		if(len < minlen || score < k) goto done;
	}
	while(bi >= 0 || bri >= 0) {
		if(ties) {
			for(int side = 0; side < 2; ++side) {
				if(!(rc & (1 << side))) continue;
				anker *V = side ? VR : VF;
				int *bt = side ? bestT_r : bestT;
				const int bidx = side ? bri : bi, vstart = side ? VR_start : VF_start;
				int v = bidx;
				while((v = tie_anker(V, start < vstart ? vstart : start, v, bidx)) >= 0) {
					if((double) (unsigned) (V[v].end - (unsigned) start) < coverT * len) v = -1;      /* (unsigned arithmetic in the reference) */
					else {
						for(int i = 1; i <= bt[0]; ++i) { c.include[bt[i]] = 1; c.Score[bt[i]] = 0; c.extend[bt[i]] = 0; }
						int *tail = bt + bt[0];
						const int keep = *tail;
						*tail = 0;
						chain_templates(&c, V, v, tail);
						bt[0] += *tail;
						*tail = keep;
					}
					if(v < 0) break;
				}
				for(int i = 1; i <= bt[0]; ++i) { c.include[bt[i]] = 0; c.Score[bt[i]] = 0; c.extend[bt[i]] = 0; }
			}
		}
		/* mrchain with mrc == 0 keeps everything (kmeranker.c:57-81) */
		if(rc) {
			seg_grow(&tree, (unsigned) start, (unsigned) (start + len));
			if(n_out >= out_cap) { n_out = -1; goto done; }
			orc_chain_rec *R = &out[n_out];
			int *bt, score, nT;
			if(rc & 1) {
				R->q_start = start; R->q_end = start + len;
				if(rc & 2) {
					int j = bestT[0];
					for(int i = 1; i <= bestT_r[0]; ++i) bestT[++j] = -bestT_r[i];
					bestT[0] += bestT_r[0];
					VF[bi].score = -VF[bi].score;
					VR[bri].score = 0;
					bestT_r[0] = 0;
				}
				bt = bestT; score = VF[bi].score; R->emit_rc = 0;
				VF[bi].score = 0;
			} else {
				R->q_start = seqlen - (int) VR[bri].end; R->q_end = seqlen - start;
				bt = bestT_r; score = VR[bri].score; R->emit_rc = 1;
				VR[bri].score = 0;
			}
			nT = bt[0];
			if(T_used + nT > T_cap) { n_out = -1; goto done; }
			memcpy(T_pool + T_used, bt + 1, (size_t) nT * sizeof(int));
			R->rc_flag = score; R->nT = nT; R->T = T_pool + T_used;
			T_used += nT;
			bt[0] = 0;
			++n_out;
		}
		/* next chain of either strand (savekmers.c:5827-5925) */
		ties = 0;
		rc = 0;
		for(int side = 0; side < 2; ++side) {
			anker *V = side ? VR : VF;
			int *bt = side ? bestT_r : bestT;
			int *bidx = side ? &bri : &bi, *head = side ? &headR : &headF;
			int *cs = side ? &cStart_r : &cStart;
			if(*bidx < 0) continue;
			int ok = 0;
			if(V[*bidx].score) {
				const int s = chain_templates(&c, V, *bidx, bt);
				if(s >= 0) {
					*cs = (int) V[s].start;
					const int cover = (int) seg_que(&tree, 0, (unsigned) *cs, V[*bidx].end);
					const int l = (int) V[*bidx].end - *cs;
					if(minlen <= l && cover <= coverT * l && mrs * l <= V[*bidx].score) ok = 1;
				}
			}
			if(ok) rc |= 1 << side;
			else V[*bidx].score = 0;
			while(*bidx >= 0 && V[*bidx].score == 0) {
				*bidx = best_anker(V, head, &ties);
				if(*bidx >= 0) {
					int good = 0;
					if(k < V[*bidx].score) {
						const int s = chain_templates(&c, V, *bidx, bt);
						if(s >= 0) {
							*cs = (int) V[s].start;
							const int cover = (int) seg_que(&tree, 0, (unsigned) *cs, V[*bidx].end);
							const int l = (int) V[*bidx].end - *cs;
							if(minlen <= l && cover <= coverT * l && mrs * l <= V[*bidx].score) good = 1;
						}
					}
					if(good) rc |= 1 << side;
					else V[*bidx].score = 0;
				}
			}
		}
		if(bi < 0 && bri < 0) break;
		if(bi >= 0 && bri >= 0) rc = choose_chain(&VF[bi], &VR[bri], cStart, cStart_r, coverT, &start, &len);
		else if(bi >= 0) { rc = 1; start = cStart; len = (int) VF[bi].end - start; }
		else { rc = 2; start = cStart_r; len = (int) VR[bri].end - start; }
	}
done:
	free(rseq); free(rN); free(VF); free(VR); free(bestT); free(bestT_r); free(c.Score); free(c.extend); free(c.include); free(tree.v);
	return n_out;
}
