// chain.hip -- stage 2 of KMA's DEFAULT mode on gfx950 (no -1t1): save_kmers_chain (savekmers.c:5127-5945) with its default
// helpers (kmeranker.c:25-30: getBestChainTemplates :83-234, pruneAnkers :372-398, getBestAnkerScore :400-431,
// getTieAnkerScore :477-492, chooseChain :512-595 with proxi 1, mrchain with mrc 0) and the segment tree of
// seqmenttree.c:25-233. SURVEY §8f F1.
//
// A read yields ANCHORS -- maximal runs of k-mer starts whose value list is the same -- on both strands, in forward read
// coordinates; the anchors of a strand are chained left to right per template (the reference's Score / extendScore / include
// arrays); chains are then taken out best first: the chain's templates, the anchors it silences, ties, the overlap with what was
// taken before (segment tree, coverT), until nothing is left. Every accepted chain is one S2 record with its query bounds, so a
// chimeric read maps in pieces.
//
// First device form: ONE LANE PER READ running the sequential algorithm on its own scratch in HBM (anchors, DB_size-wide
// per-template arrays, template lists, tree nodes); reads are handed out by static strides. What is parallel is the reads.
// Correct first (records identical to the reference's -s2 tap); the run-of-equal-lists walk of scan.hip is the obvious next step
// for the anchor search, which is where the gathers are.
#include "kmahip_internal.h"
#include "dna_dev.h"
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

constexpr uint32_t NOLIST = 0xFFFFFFFFu;

struct CAnk { int score, weight, score_len, len_len; unsigned start, end; uint32_t values; int descend; };
struct CSeg { unsigned start, end, covered; int b0, b1; };

struct ChainArgs {
	DevDB db;
	int64_t n_reads;
	const uint64_t *seq;
	const int64_t *seq_off;
	const int32_t *len;
	const int32_t *N;
	const int64_t *N_off;
	int M, MM, U, W1, Wl;
	int exhaustive, minlen;
	int stop_after;              // timing experiments only (KMAHIP_CHAIN_STOP): 1 = after the anchors, 2 = after the chaining
	double coverT, mrs;
	// scratch, one region per lane
	uint8_t *scratch;
	int64_t lane_bytes, lanes;
	int a_cap, b_cap, s_cap;     // anchors per strand, template-list slots, tree nodes
	// out
	int32_t *rec;                // 8 ints per record: read lo, read hi, ordinal, rc_flag, emit_rc, q_start, q_end, nT
	int64_t *rec_T;              // first template of the record in `T`
	int32_t *T;
	int64_t rec_cap, T_cap;
	unsigned long long *counters;   // [0] records, [1] status, [2] templates
};

struct CLane {
	const DevDB *db;
	CAnk *VF, *VR;
	int *Score, *extend;
	int8_t *include;
	int *bestT, *bestT_r;
	CSeg *tree;
	int tree_n;
	int k, M, MM, U, W1, Wl;
	int a_cap, b_cap, s_cap;
	int status;                  // 1: a per-lane capacity ran out
};

__device__ __forceinline__ uint32_t db_probe(const DevDB &db, uint32_t key) {
	const uint32_t sh = 32u - db.nb_log2, nbm = (1u << db.nb_log2) - 1u;
	uint32_t b = (key * 0x9E3779B1u) >> sh;
	for(;;) {
		const uint4 *p = reinterpret_cast<const uint4 *>(db.slots + (size_t) b * KMAHIP_BUCKET_SLOTS);
		const uint4 a = p[0], c = p[1];
		if(a.x == key && a.y != KMAHIP_EMPTY_VI) return a.y;
		if(a.z == key && a.w != KMAHIP_EMPTY_VI) return a.w;
		if(c.x == key && c.y != KMAHIP_EMPTY_VI) return c.y;
		if(c.z == key && c.w != KMAHIP_EMPTY_VI) return c.w;
		if(c.w == KMAHIP_EMPTY_VI) return NOLIST;
		b = (b + 1u) & nbm;
	}
}
// value list of a k-mer: its offset, or NOLIST
__device__ __forceinline__ uint32_t list_of(const DevDB &db, uint32_t key) {
	const uint32_t gp = db_probe(db, key);
	return gp == NOLIST ? NOLIST : db.vs_id[gp];
}
__device__ __forceinline__ int list_n(const DevDB &db, uint32_t v) { return db.values_u16 ? (int) db.values16[v] : (int) db.values32[v]; }
__device__ __forceinline__ int list_at(const DevDB &db, uint32_t v, int i) { return db.values_u16 ? (int) db.values16[v + i] : (int) db.values32[v + i]; }

// k bases of the reverse-complemented read from position pos, zeros behind its end (the reference's buffer, freshly cleared)
__device__ __forceinline__ uint32_t rc_kmer(const QView &qr, int pos, int k) {
	if(pos < 0 || pos >= qr.L) return 0u;
	uint64_t w = qwin(qr, pos);
	const int have = qr.L - pos;
	if(have < 32) w &= ~0ull << (64 - 2 * have);
	return (uint32_t) (w >> (64 - 2 * k));
}

// ---- segment tree (seqmenttree.c), recursion unrolled by depth -------------------------------------------------------------
template <int D> struct SegOps {
	__device__ static unsigned add(CLane &L, int root, int node) {
		CSeg *v = L.tree;
		CSeg &R = v[root], &Nn = v[node];
		if(R.b0 >= 0) {
			if(Nn.start < R.start && R.end < Nn.end) {
				R.start = Nn.start; R.end = Nn.end; R.covered = Nn.covered; Nn.covered = 0; R.b0 = -1;
				return R.covered;
			} else if(R.end < Nn.end) R.end = Nn.end;
			else if(Nn.start < R.start) R.start = Nn.start;
			unsigned pos = v[R.b1].start;
			if(Nn.end < pos) R.covered = v[R.b1].covered + SegOps<D - 1>::add(L, R.b0, node);
			else if(pos <= Nn.start) R.covered = v[R.b0].covered + SegOps<D - 1>::add(L, R.b1, node);
			else {
				pos = Nn.start;
				Nn.start = v[R.b0].end + 1;
				Nn.covered = Nn.end - Nn.start;
				const unsigned covered = SegOps<D - 1>::add(L, R.b1, node);
				Nn.start = pos;
				Nn.end = v[R.b0].end;
				Nn.covered = Nn.end - Nn.start;
				R.covered = covered + SegOps<D - 1>::add(L, R.b0, node);
			}
		} else if(Nn.end < R.start || R.end < Nn.start) {
			const int bud = node + 1;
			v[bud].start = R.start; v[bud].end = R.end; v[bud].covered = R.covered; v[bud].b0 = -1;
			if(Nn.end < R.start) { R.start = Nn.start; R.b0 = node; R.b1 = bud; }
			else { R.end = Nn.end; R.b0 = bud; R.b1 = node; }
			R.covered += Nn.covered;
		} else {
			if(Nn.start < R.start) R.start = Nn.start;
			if(R.end < Nn.end) R.end = Nn.end;
			Nn.covered = 0;
			R.covered = R.end - R.start;
		}
		return R.covered;
	}
	__device__ static unsigned que(const CLane &L, int i, unsigned start, unsigned end) {
		const CSeg &s = L.tree[i];
		if(end < s.start || s.end < start) return 0;
		if(start <= s.start && s.end <= end) return s.covered;
		if(s.b0 >= 0) return SegOps<D - 1>::que(L, s.b0, start, end) + SegOps<D - 1>::que(L, s.b1, start, end);
		if(s.start <= start && end <= s.end) return end - start;
		if(s.start <= start && start < s.end) return s.end - start;
		if(s.start < end && end <= s.end) return end - s.start;
		return 0;
	}
};
template <> struct SegOps<0> {
	__device__ static unsigned add(CLane &L, int, int) { L.status = 1; return 0; }
	__device__ static unsigned que(const CLane &, int, unsigned, unsigned) { return 0; }
};
constexpr int SEG_DEPTH = 24;

__device__ void seg_grow(CLane &L, unsigned start, unsigned end) {
	if(L.s_cap <= L.tree_n + 2) { L.status = 1; return; }
	CSeg *v = L.tree;
	if(L.tree_n == 0) {
		L.tree_n = 1;
		v[0].start = start; v[0].end = end; v[0].covered = end - start; v[0].b0 = v[0].b1 = -1;
		return;
	}
	const int node = L.tree_n;
	v[node].start = start; v[node].end = end; v[node].covered = end - start; v[node].b0 = -1;
	v[0].covered = SegOps<SEG_DEPTH>::add(L, 0, node);
	if(v[node].covered) L.tree_n += 2;
}

// ---- chaining helpers -------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bridge(const CLane &L, int mlen, int weight, int gaps) {
	const int k = L.k, M = L.M, MM = L.MM, U = L.U, W1 = L.W1;
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

// getBestChainTemplates: templates of the chain that ends in V[src] into bests[0 .. ]; the anchors it passes are silenced.
// Returns the anchor the chain starts at, -1 none. `room`: slots bests may use.
__device__ int chain_templates(CLane &L, CAnk *V, int src, int *bests, int room) {
	const DevDB &db = *L.db;
	if(src < 0) return -1;
	int nextAnker = 0;
	{
		const int n = list_n(db, V[src].values);
		if(n + 1 > room) { L.status = 1; bests[0] = 0; return -1; }
		bests[0] = n;
		for(int i = n; i >= 1; --i) {
			const int t = list_at(db, V[src].values, i);
			bests[i] = t;
			if(++L.include[t] == 1) nextAnker = 1;
		}
	}
	const int bestScore = V[src].score;
	int prev = src;
	for(int node = src; nextAnker && node >= 0; --node) {
		const int n = list_n(db, V[node].values);
		const int start = (int) V[node].start, end = (int) V[node].end;
		for(int i = n; i >= 1; --i) {
			const int t = list_at(db, V[node].values, i);
			if(!L.include[t]) continue;
			int score = L.Score[t];
			const int pos = L.extend[t];
			if(pos == 0) score = V[node].weight;
			else {
				score += bridge(L, (int) db.mlen, V[node].weight, pos - end);
				V[node].score = 0;
			}
			if(bestScore <= score) {
				int tmp;
				if(V[node].start) {
					tmp = L.W1 + ((int) V[node].start - 1) * L.U;
					tmp = score + (L.Wl < tmp ? tmp : L.Wl);
				} else tmp = score;
				if(tmp == bestScore) { score = bestScore; nextAnker = 0; prev = node; }
			}
			L.extend[t] = start;
			L.Score[t] = score;
		}
	}
	int j = 0;
	for(int i = 1; i <= bests[0]; ++i) {
		const int t = bests[i];
		if(L.include[t] == 1 && bestScore <= L.Score[t]) bests[++j] = t;
		L.Score[t] = 0; L.include[t] = 0; L.extend[t] = 0;
	}
	bests[0] = j;
	return j ? prev : -1;
}

__device__ int prune(CAnk *V, int head, int k) {
	while(head >= 0 && V[head].score < k) head = V[head].descend;
	if(head < 0) return -1;
	int prev = head;
	for(int node = V[head].descend; node >= 0; node = V[node].descend) if(k <= V[node].score) { V[prev].descend = node; prev = node; }
	V[prev].descend = -1;
	return head;
}

__device__ int best_anker(CAnk *V, int *head, unsigned *ties) {
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

__device__ int tie_anker(const CAnk *V, int stop, int src, int best) {
	if(src < 0 || (int) V[src].start <= stop) return -1;
	while(src > 0 && stop < (int) V[--src].start) if(V[src].score == V[best].score) return src;
	return -1;
}

__device__ int choose_chain(const CAnk &b, const CAnk &r, int cStart, int cStart_r, double coverT, int *Start, int *Len) {
	int rc = r.score < b.score ? 1 : b.score < r.score ? 2 : 3, start, end;
	if(rc == 1) { start = cStart; end = (int) b.end; }
	else if(rc == 2) { start = cStart_r; end = (int) r.end; }
	else if((int) b.end < cStart_r) { start = cStart; end = (int) b.end; rc = 1; }
	else if((int) r.end < cStart) { start = cStart_r; end = (int) r.end; rc = 2; }
	else if(cStart <= cStart_r && r.end <= b.end) { start = cStart; end = (int) b.end; }
	else if(cStart_r <= cStart && b.end <= r.end) { start = cStart_r; end = (int) r.end; }
	else if(r.end < b.end) {
		start = (int) b.end - cStart;
		end = (int) r.end - cStart_r;
		end = start < end ? start : end;
		start = cStart_r;
		if(coverT * end <= (int) r.end - cStart) end = (int) b.end;
		else { end = (int) r.end; rc = 2; }
	} else {
		start = (int) b.end - cStart;
		end = (int) r.end - cStart_r;
		end = start < end ? start : end;
		start = cStart;
		if(coverT * end <= (int) b.end - cStart_r) end = (int) r.end;
		else { end = (int) b.end; rc = 1; }
	}
	*Start = start; *Len = end - start;
	return rc;
}

// anchors of one strand in forward coordinates (savekmers.c:5208-5330, 5333-5452); returns their number
__device__ int build_ankers(CLane &L, const QView &qf, const QView &qr, int exhaustive, int is_rc, CAnk *V) {
	const DevDB &db = *L.db;
	const int k = L.k, seqlen = qf.L, nN = qf.nN;
	V[0].start = 0; V[0].end = 0; V[0].values = NOLIST; V[0].descend = -1;
	bool HIT = exhaustive != 0;
	{
		const QView &q = is_rc ? qr : qf;
		int j = 0;
		for(int i = 1; i <= nN + 1 && !HIT; ++i) {
			const int segend = i <= nN ? qN_at(q, i) : seqlen;
			for(; j < segend - k + 1 && !HIT; j += k) if(db_probe(db, is_rc ? rc_kmer(qr, j, k) : q_kmer(qf, j, k)) != NOLIST) HIT = true;
			j = segend + 1;
		}
	}
	if(!HIT) return 0;
	int hits = 0, v = 0, Ms = 0, MMs = 0, gaps = 0, j = 0;
	uint32_t last = NOLIST;
	const int seqend = seqlen - k + 1;
	// behind an N the reference restarts the reverse strand's k-mer at seqlen - j, k bases further on than seqlen - k - j
	// (savekmers.c:5447-5449); kept as it is
	int rcpos = seqlen - k;
	for(int i = 1; i <= nN + 1 && j < seqend; ++i) {
		const int segend = i <= nN ? qN_at(qf, i) : seqlen;
		uint32_t gp = NOLIST;                  // where the k-mer of the step before lies in the template store
		for(; j < segend - k + 1; ++j, --rcpos) {
			// A read that matches a template keeps matching it: when the base that enters the window is the template's next base
			// (the one in front of it on the reverse strand, whose windows move backwards), the k-mer is the template's
			// neighbouring k-mer and its value list stands in vs_id -- no probe (the walk of scan.hip, one lane here)
			uint32_t values;
			bool walked = false;
			if(gp != NOLIST) {
				if(!is_rc) {
					const uint32_t v = db.vs_id[gp + 1];
					if(v != KMAHIP_EMPTY_VI && (int) ((db.cat[(gp + k) >> 5] >> (62 - (((gp + k) & 31) << 1))) & 3ull) == q2(qf, j + k - 1)) { ++gp; values = v; walked = true; }
				} else if(gp > 0 && rcpos >= 0 && rcpos + k <= seqlen) {
					const uint32_t v = db.vs_id[gp - 1];
					if(v != KMAHIP_EMPTY_VI && (int) ((db.cat[(gp - 1) >> 5] >> (62 - (((gp - 1) & 31) << 1))) & 3ull) == q2(qr, rcpos)) { --gp; values = v; walked = true; }
				}
			}
			if(!walked) {
				gp = db_probe(db, is_rc ? rc_kmer(qr, rcpos, k) : q_kmer(qf, j, k));
				values = gp == NOLIST ? NOLIST : db.vs_id[gp];
				if(is_rc && (rcpos < 0 || rcpos + k > seqlen)) gp = NOLIST;       // (a window that hangs over the end: nothing to walk from)
			}
			if(values != NOLIST) {
				bool open = true;
				if(values == last) {
					if(gaps == 0) { ++Ms; open = false; }
					else if(gaps == k) { Ms += k; ++MMs; open = false; }
				}
				if(open) {
					if(last != NOLIST) {
						V[v].weight = Ms * L.M + MMs * L.MM;
						V[v].end = (unsigned) (j - gaps + k);
						V[v].descend = v + 1;
						++v;
						if(v >= L.a_cap) { L.status = 1; return 0; }
					}
					V[v].start = (unsigned) j; V[v].values = values; V[v].descend = -1;
					last = values;
					Ms = k; MMs = 0;
					++hits;
				}
				gaps = 0;
			} else ++gaps;
		}
		gaps += segend + 1 - j;
		j = segend + 1;
		rcpos = seqlen - j;
	}
	if(last != NOLIST) {
		V[v].weight = Ms * L.M + MMs * L.MM;
		V[v].end = (unsigned) (seqlen - gaps);
	}
	return hits;
}

struct Emit {
	const ChainArgs *A;
	int64_t read;
	int ordinal;
};

__device__ void emit_record(CLane &L, Emit &E, int rc_flag, int emit_rc, int q_start, int q_end, const int *bt) {
	const ChainArgs &A = *E.A;
	const int nT = bt[0];
	const unsigned long long slot = atomicAdd(&A.counters[0], 1ull);
	const unsigned long long toff = atomicAdd(&A.counters[2], (unsigned long long) nT);
	if((int64_t) slot >= A.rec_cap || (int64_t) (toff + nT) > A.T_cap) { atomicMax(&A.counters[1], 2ull); ++E.ordinal; return; }
	int32_t *r = A.rec + 8 * slot;
	r[0] = (int32_t) (E.read & 0xFFFFFFFFll); r[1] = (int32_t) (E.read >> 32); r[2] = E.ordinal++; r[3] = rc_flag; r[4] = emit_rc;
	r[5] = q_start; r[6] = q_end; r[7] = nT;
	A.rec_T[slot] = (int64_t) toff;
	for(int i = 0; i < nT; ++i) A.T[toff + i] = bt[1 + i];
}

__device__ void chain_read(CLane &L, const ChainArgs &A, int64_t r) {
	const DevDB &db = *L.db;
	const int k = L.k;
	QView qf;
	qf.w = A.seq + A.seq_off[r]; qf.L = A.len[r]; qf.N = A.N + A.N_off[r]; qf.nN = (int) (A.N_off[r + 1] - A.N_off[r]); qf.rc = 0;
	QView qr = qf; qr.rc = 1;
	const int seqlen = qf.L;
	if(seqlen < k) return;
	if(seqlen + 2 > L.a_cap) { L.status = 1; return; }
	CAnk *VF = L.VF, *VR = L.VR;
	int *bestT = L.bestT, *bestT_r = L.bestT_r;
	L.tree_n = 0;
	Emit E = {&A, r, 0};

	const unsigned hitF = (unsigned) build_ankers(L, qf, qr, A.exhaustive, 0, VF);
	const unsigned hitR = (unsigned) build_ankers(L, qf, qr, A.exhaustive, 1, VR);
	if(L.status || (!hitF && !hitR)) return;
	if(A.stop_after == 1) return;

	// chains left to right, per strand (savekmers.c:5466-5634)
	CAnk *best = nullptr, *best_r = &VF[0];
	unsigned ties = 0;
	VF[0].score = 0;
	{
		int *bests = bestT;
		bestT[0] = 0; bestT_r[0] = 0;
		for(int strand = 0; strand < 2; ++strand) {
			CAnk *V = strand ? VR : VF;
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
				CAnk &An = V[vi];
				const int start = (int) An.start, end = (int) An.end;
				An.score = 0; An.score_len = 0; An.len_len = 1;
				const int n = list_n(db, An.values);
				for(int i = n; i >= 1; --i) {
					const int t = list_at(db, An.values, i);
					int score = L.Score[t];
					const int pos = L.extend[t];
					if(!L.include[t]) {
						L.include[t] = 1;
						if(bests[0] + 2 > L.b_cap) { L.status = 1; return; }
						bests[++bests[0]] = t;
						if(start) {
							score = L.W1 + (start - 1) * L.U;
							score = An.weight + (L.Wl < score ? score : L.Wl);
						} else score = An.weight;
					} else {
						score += bridge(L, (int) db.mlen, An.weight, start - pos);
						if(score < 0) {
							int test = start ? L.W1 + (start - 1) * L.U : 0;
							if(test < L.Wl) test = L.Wl;
							if(score < test + An.weight) score = test + An.weight;
						}
					}
					if(An.score < score) An.score = score;
					int len_len = db.tlen[t];
					if(seqlen < len_len) len_len = seqlen;
					double score_len = score;
					if(An.len_len != len_len) { score_len /= len_len; score_len *= An.len_len; }
					if(An.score_len < score_len || (An.score_len == score_len && An.score_len < score)) { An.score_len = score; An.len_len = len_len; }
					L.Score[t] = score;
					L.extend[t] = end;
				}
				if(best_r->score < An.score) { best_r = &An; ties = 0; }
				else if(best_r->score == An.score) {
					if(best_r->score_len < An.score_len) { best_r = &An; ties = 0; }
					else { best_r = &An; ++ties; }
				}
				++vi;
			}
			for(int i = 1; i <= bests[0]; ++i) { const int t = bests[i]; L.Score[t] = 0; L.extend[t] = 0; L.include[t] = 0; }
		}
	}
	if(best->score < k && best_r->score < k) return;
	if(A.stop_after == 2) return;

	const int VF_start = (int) VF[0].start, VR_start = (int) VR[0].start;
	int headF = prune(VF, 0, k), headR = prune(VR, 0, k);
	if(headF < 0) best->score = 0;
	if(headR < 0) best_r->score = 0;
	bestT[0] = 0; bestT_r[0] = 0;
	int bi = (int) (best - VF), bri = (int) (best_r - VR);
	int cStart = -1, cStart_r = -1, start = 0, len = 0, rc;
	if(!best->score || !best_r->score) {
		if(best->score) {
			const int s = chain_templates(L, VF, bi, bestT, L.b_cap);
			if(s < 0) return;
			cStart = (int) VF[s].start; start = cStart; len = (int) VF[bi].end - start; rc = 1;
		} else {
			const int s = chain_templates(L, VR, bri, bestT_r, L.b_cap);
			if(s < 0) return;
			cStart_r = (int) VR[s].start; start = cStart_r; len = (int) VR[bri].end - start; rc = 2;
		}
	} else {
		int s = chain_templates(L, VF, bi, bestT, L.b_cap);
		if(s < 0) return;
		cStart = (int) VF[s].start;
		s = chain_templates(L, VR, bri, bestT_r, L.b_cap);
		if(s < 0) return;
		cStart_r = (int) VR[s].start;
		rc = choose_chain(VF[bi], VR[bri], cStart, cStart_r, A.coverT, &start, &len);
	}
	{
		const int score = VF[bi].score > VR[bri].score ? VF[bi].score : VR[bri].score;
		if(len < A.minlen || score < k) return;
	}
	while((bi >= 0 || bri >= 0) && !L.status) {
		if(ties) {
			for(int side = 0; side < 2; ++side) {
				if(!(rc & (1 << side))) continue;
				CAnk *V = side ? VR : VF;
				int *bt = side ? bestT_r : bestT;
				const int bidx = side ? bri : bi, vstart = side ? VR_start : VF_start;
				int v = bidx;
				while((v = tie_anker(V, start < vstart ? vstart : start, v, bidx)) >= 0) {
					if((double) (unsigned) (V[v].end - (unsigned) start) < A.coverT * len) break;      // (unsigned arithmetic in the reference)
					for(int i = 1; i <= bt[0]; ++i) { const int t = bt[i]; L.include[t] = 1; L.Score[t] = 0; L.extend[t] = 0; }
					int *tail = bt + bt[0];
					const int keep = *tail;
					*tail = 0;
					chain_templates(L, V, v, tail, L.b_cap - bt[0]);
					bt[0] += *tail;
					*tail = keep;
					if(L.status) return;
				}
				for(int i = 1; i <= bt[0]; ++i) { const int t = bt[i]; L.include[t] = 0; L.Score[t] = 0; L.extend[t] = 0; }
			}
		}
		if(rc) {
			seg_grow(L, (unsigned) start, (unsigned) (start + len));
			if(L.status) return;
			if(rc & 1) {
				if(rc & 2) {
					int j = bestT[0];
					if(j + bestT_r[0] + 2 > L.b_cap) { L.status = 1; return; }
					for(int i = 1; i <= bestT_r[0]; ++i) bestT[++j] = -bestT_r[i];
					bestT[0] += bestT_r[0];
					VF[bi].score = -VF[bi].score;
					VR[bri].score = 0;
					bestT_r[0] = 0;
				}
				emit_record(L, E, VF[bi].score, 0, start, start + len, bestT);
				VF[bi].score = 0;
				bestT[0] = 0;
			} else {
				emit_record(L, E, VR[bri].score, 1, seqlen - (int) VR[bri].end, seqlen - start, bestT_r);
				VR[bri].score = 0;
				bestT_r[0] = 0;
			}
		}
		// next chain of either strand (savekmers.c:5827-5925)
		ties = 0;
		rc = 0;
		for(int side = 0; side < 2; ++side) {
			CAnk *V = side ? VR : VF;
			int *bt = side ? bestT_r : bestT;
			int &bidx = side ? bri : bi, &head = side ? headR : headF, &cs = side ? cStart_r : cStart;
			if(bidx < 0) continue;
			bool ok = false;
			if(V[bidx].score) {
				const int s = chain_templates(L, V, bidx, bt, L.b_cap);
				if(s >= 0) {
					cs = (int) V[s].start;
					const int cover = (int) SegOps<SEG_DEPTH>::que(L, 0, (unsigned) cs, V[bidx].end);
					const int l = (int) V[bidx].end - cs;
					ok = A.minlen <= l && cover <= A.coverT * l && A.mrs * l <= V[bidx].score;
				}
			}
			if(ok) rc |= 1 << side;
			else V[bidx].score = 0;
			while(bidx >= 0 && V[bidx].score == 0 && !L.status) {
				bidx = best_anker(V, &head, &ties);
				if(bidx >= 0) {
					bool good = false;
					if(k < V[bidx].score) {
						const int s = chain_templates(L, V, bidx, bt, L.b_cap);
						if(s >= 0) {
							cs = (int) V[s].start;
							const int cover = (int) SegOps<SEG_DEPTH>::que(L, 0, (unsigned) cs, V[bidx].end);
							const int l = (int) V[bidx].end - cs;
							good = A.minlen <= l && cover <= A.coverT * l && A.mrs * l <= V[bidx].score;
						}
					}
					if(good) rc |= 1 << side;
					else V[bidx].score = 0;
				}
			}
		}
		if(bi < 0 && bri < 0) break;
		if(bi >= 0 && bri >= 0) rc = choose_chain(VF[bi], VR[bri], cStart, cStart_r, A.coverT, &start, &len);
		else if(bi >= 0) { rc = 1; start = cStart; len = (int) VF[bi].end - start; }
		else { rc = 2; start = cStart_r; len = (int) VR[bri].end - start; }
	}
}

#ifndef CHAIN_MIN_WAVES
#define CHAIN_MIN_WAVES 2
#endif
__global__ __launch_bounds__(64, CHAIN_MIN_WAVES) void chain_kernel(const ChainArgs A) {
	const int64_t lane = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(lane >= A.lanes) return;
	uint8_t *base = A.scratch + lane * A.lane_bytes;
	const int64_t D = A.db.DB_size;
	CLane L;
	L.db = &A.db;
	L.VF = (CAnk *) base; base += (size_t) A.a_cap * sizeof(CAnk);
	L.VR = (CAnk *) base; base += (size_t) A.a_cap * sizeof(CAnk);
	L.Score = (int *) base; base += (size_t) (D + 1) * 4;
	L.extend = (int *) base; base += (size_t) (D + 1) * 4;
	L.bestT = (int *) base; base += (size_t) A.b_cap * 4;
	L.bestT_r = (int *) base; base += (size_t) A.b_cap * 4;
	L.tree = (CSeg *) base; base += (size_t) A.s_cap * sizeof(CSeg);
	L.include = (int8_t *) base;
	L.k = (int) A.db.kmersize; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1; L.Wl = A.Wl;
	L.a_cap = A.a_cap; L.b_cap = A.b_cap; L.s_cap = A.s_cap;
	L.status = 0; L.tree_n = 0;
	for(int64_t r = lane; r < A.n_reads; r += A.lanes) {
		chain_read(L, A, r);
		if(L.status) {
			// leave the per-template arrays clean for the next read of this lane, and say so
			atomicMax(&A.counters[1], 40ull);
			for(int64_t t = 0; t <= D; ++t) { L.Score[t] = 0; L.extend[t] = 0; L.include[t] = 0; }
			L.status = 0;
		}
	}
}

}  // namespace

// ---- the launch, everything in HBM: `d` holds DEVICE pointers; rec (8 ints per record: read lo, read hi, ordinal within the read,
// rc_flag, emit_rc, q_start, q_end, number of templates), rec_T (first template of the record in T) and T are device buffers of
// rec_cap / T_cap entries, filled in no particular order. n_recs / n_T: what the batch needs (KMAHIP_EOVERFLOW when that is more). -----
int kmahip_chain_device(kmahip_db *db, const kmahip_reads *d, const kmahip_params *p, const kmahip_chain_params *cp, int32_t *rec, int64_t *rec_T,
                        int32_t *T, int64_t rec_cap, int64_t T_cap, int64_t *n_recs, int64_t *n_T) {
	const int64_t n = d->n_reads;
	*n_recs = 0; *n_T = 0;
	if(n <= 0) return KMAHIP_OK;
	if(!db->dev.tlen) { kmahip_set_error("index has no .length.b: the default template finder needs the template lengths"); return KMAHIP_EINVAL; }
	const int max_len = d->max_len;
	if(max_len <= 0) { kmahip_set_error("kmahip_reads.max_len must be set"); return KMAHIP_EINVAL; }
	const int64_t D = db->info.DB_size;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto stamp = [&](const char *what) {
		if(!dbg) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[kmahip] scan_chain: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	ChainArgs A;
	A.db = db->dev; A.n_reads = n;
	A.seq = d->seq; A.seq_off = d->seq_off; A.len = d->len; A.N = d->N; A.N_off = d->N_off;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1; A.Wl = p->rw.Wl;
	A.stop_after = getenv("KMAHIP_CHAIN_STOP") ? atoi(getenv("KMAHIP_CHAIN_STOP")) : 0;
	A.exhaustive = p->exhaustive; A.minlen = cp ? cp->minlen : 16; A.coverT = cp ? cp->coverT : 0.1; A.mrs = cp ? cp->mrs : 0.5;
	// (tree nodes: two per accepted chain, and a chain is minlen bases at least; 128 = 63 chains cover every read up to 1 kb)
	A.a_cap = max_len + 4; A.b_cap = (int) std::min<int64_t>(2 * D + 4, 2048); A.s_cap = std::max(128, 2 * (max_len / std::max(A.minlen, 8)) + 8);
	A.lane_bytes = ((int64_t) 2 * A.a_cap * (int64_t) sizeof(CAnk) + (D + 1) * 8 + (int64_t) 2 * A.b_cap * 4 + (int64_t) A.s_cap * (int64_t) sizeof(CSeg) + (D + 1) + 63) & ~63ll;
	// 254 VGPRs: one wave per SIMD = 65 536 lanes resident. (Capped at 128 VGPRs for four waves per SIMD the kernel spills 1 000
	// registers and takes as long: 2 M reads in 62 vs 68 ms. At ~2 000 scattered accesses per read that is ~65 G lines/s, the
	// gather ceiling of DESIGN 3.1 -- the way up is fewer scattered accesses, anchors and lists out of HBM scratch, not more lanes.)
	int64_t lanes = getenv("KMAHIP_CHAIN_LANES") ? atoll(getenv("KMAHIP_CHAIN_LANES")) : 65536;
	while(lanes > 64 && lanes * A.lane_bytes > (16ll << 30)) lanes >>= 1;
	lanes = std::min<int64_t>(lanes, ((n + 63) / 64) * 64);
	A.lanes = lanes;
	void *scratch = nullptr, *counters = nullptr;
	struct Free { void *&a, *&b; ~Free() { if(a) (void) hipFree(a); if(b) (void) hipFree(b); } } guard{scratch, counters};
	if(hipMalloc(&scratch, (size_t) (lanes * A.lane_bytes)) != hipSuccess) { scratch = nullptr; kmahip_set_error("hipMalloc of %lld bytes failed", (long long) (lanes * A.lane_bytes)); return KMAHIP_ENOMEM; }
	if(hipMalloc(&counters, KMAHIP_N_COUNTERS * 8) != hipSuccess) { counters = nullptr; kmahip_set_error("hipMalloc failed"); return KMAHIP_ENOMEM; }
	HIP_TRY(hipMemsetAsync(scratch, 0, (size_t) (lanes * A.lane_bytes), 0));
	HIP_TRY(hipMemsetAsync(counters, 0, KMAHIP_N_COUNTERS * 8, 0));
	A.scratch = (uint8_t *) scratch; A.counters = (unsigned long long *) counters;
	A.rec = rec; A.rec_T = rec_T; A.T = T; A.rec_cap = rec_cap; A.T_cap = T_cap;
	if(dbg) { HIP_TRY(hipDeviceSynchronize()); stamp("scratch allocated and cleared"); }
	hipLaunchKernelGGL(chain_kernel, dim3((unsigned) (lanes / 64)), dim3(64), 0, 0, A);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipDeviceSynchronize());
	stamp("chain_kernel");
	unsigned long long c[3] = {0, 0, 0};
	HIP_TRY(hipMemcpy(c, A.counters, sizeof c, hipMemcpyDeviceToHost));
	*n_recs = (int64_t) c[0]; *n_T = (int64_t) c[2];
	if(c[1] == 40) { kmahip_set_error("default template finder: a per-read capacity ran out (value lists of more than %d templates, more than %d chains in a read, or chains nested deeper than %d in the tree of covered stretches)", A.b_cap / 2, A.s_cap / 2, SEG_DEPTH); return KMAHIP_EOVERFLOW; }
	if(c[1] == 2 || (int64_t) c[0] > rec_cap || (int64_t) c[2] > T_cap) { kmahip_set_error("record capacity: %llu records with %llu templates", c[0], c[2]); return KMAHIP_EOVERFLOW; }
	return KMAHIP_OK;
}

// ---- C-ABI ------------------------------------------------------------------------------------------------------------------
extern "C" int kmahip_scan_chain(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                                 const kmahip_chain_params *cp, kmahip_chain_recs *out) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	out->n_recs = 0; out->n_T = 0;
	if(n <= 0) return KMAHIP_OK;
	if(reads->max_len <= 0) { kmahip_set_error("kmahip_reads.max_len must be set"); return KMAHIP_EINVAL; }
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto t_last = std::chrono::steady_clock::now();
	auto stamp = [&](const char *what) {
		if(!dbg) return;
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[kmahip] scan_chain: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	// device buffers: staged reads, outputs
	std::vector<void *> owned;
	struct Free { std::vector<void *> &v; ~Free() { for(void *q : v) (void) hipFree(q); } } guard{owned};
	auto dev = [&](size_t bytes, void **d, const void *src, bool zero) -> int {
		if(hipMalloc(d, bytes ? bytes : 16) != hipSuccess) { kmahip_set_error("hipMalloc of %zu bytes failed", bytes); return KMAHIP_ENOMEM; }
		owned.push_back(*d);
		if(zero && hipMemset(*d, 0, bytes ? bytes : 16) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
		if(src && bytes && hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
		return KMAHIP_OK;
	};
	int rc;
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;
	{
		std::vector<uint64_t> seq((size_t) reads->seq_words + 2, 0);
		if(reads->seq_words) memcpy(seq.data(), reads->seq, (size_t) reads->seq_words * 8);
		if((rc = dev(seq.size() * 8, (void **) &d.seq, seq.data(), false))) return rc;
	}
	int32_t *d_rec, *d_T;
	int64_t *d_rec_T;
	if((rc = dev((size_t) (n + 1) * 8, (void **) &d.seq_off, reads->seq_off, false)) || (rc = dev((size_t) n * 4, (void **) &d.len, reads->len, false)) ||
	   (rc = dev((size_t) std::max<int64_t>(reads->N_total, 1) * 4, (void **) &d.N, reads->N_total ? reads->N : nullptr, !reads->N_total)) ||
	   (rc = dev((size_t) (n + 1) * 8, (void **) &d.N_off, reads->N_off, false)) ||
	   (rc = dev((size_t) std::max<int64_t>(out->rec_cap, 1) * 32, (void **) &d_rec, nullptr, false)) ||
	   (rc = dev((size_t) std::max<int64_t>(out->rec_cap, 1) * 8, (void **) &d_rec_T, nullptr, false)) ||
	   (rc = dev((size_t) std::max<int64_t>(out->T_cap, 1) * 4, (void **) &d_T, nullptr, false))) return rc;
	stamp("reads staged");
	if((rc = kmahip_chain_device(db, &d, p, cp, d_rec, d_rec_T, d_T, out->rec_cap, out->T_cap, &out->n_recs, &out->n_T))) return rc;
	unsigned long long c[3] = {(unsigned long long) out->n_recs, 0, (unsigned long long) out->n_T};
	struct { int32_t *rec; int64_t *rec_T; int32_t *T; } A = {d_rec, d_rec_T, d_T};
	t_last = std::chrono::steady_clock::now();
	// back to the host, in stream order (reads ascending, a read's chains in the order they were taken)
	const size_t m = (size_t) c[0];
	std::vector<int32_t> rec(m * 8 + 8);
	std::vector<int64_t> rT(m + 1);
	std::vector<int32_t> T((size_t) c[2] + 1);
	if(m) { HIP_TRY(hipMemcpy(rec.data(), A.rec, m * 32, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(rT.data(), A.rec_T, m * 8, hipMemcpyDeviceToHost)); }
	if(c[2]) HIP_TRY(hipMemcpy(T.data(), A.T, (size_t) c[2] * 4, hipMemcpyDeviceToHost));
	// stream order: reads ascending, a read's chains by their ordinal (0, 1, ... per read) -- a counting sort
	auto read_of = [&](size_t x) { return (int64_t) (uint32_t) rec[8 * x] | ((int64_t) rec[8 * x + 1] << 32); };
	std::vector<size_t> order(m);
	{
		std::vector<int64_t> first((size_t) n + 1, 0);
		for(size_t x = 0; x < m; ++x) ++first[(size_t) read_of(x) + 1];
		for(int64_t r = 0; r < n; ++r) first[(size_t) r + 1] += first[(size_t) r];
		for(size_t x = 0; x < m; ++x) order[(size_t) (first[(size_t) read_of(x)] + rec[8 * x + 2])] = x;
	}
	int64_t at = 0;
	for(size_t x = 0; x < m; ++x) {
		const size_t s = order[x];
		out->read[x] = read_of(s); out->rc_flag[x] = rec[8 * s + 3]; out->emit_rc[x] = rec[8 * s + 4];
		out->q_start[x] = rec[8 * s + 5]; out->q_end[x] = rec[8 * s + 6];
		out->T_off[x] = at;
		const int nT = rec[8 * s + 7];
		memcpy(out->T + at, T.data() + rT[s], (size_t) nT * 4);
		at += nT;
	}
	out->T_off[m] = at;
	stamp("records back and in stream order");
	return KMAHIP_OK;
}
