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
#include <cstring>
#include <rocprim/rocprim.hpp>
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

using CAnk = KmaAnk;      // (kmahip_internal.h: the anchors of short N-free reads come from chain_anchor_kernel, scan.hip)
struct CSeg { unsigned start, end, covered; int b0, b1; };

// What a lane keeps in HBM -- its anchors, its template lists, its tree of covered stretches, the database's arrays, the output columns --
// is reached through pointers of address space 1. As generic pointers (what a pointer kept in a struct is to the compiler) every access
// was a FLAT one: counted on two counters at once, so each was waited for before anything else went on, and a lane never had two loads
// in flight; GLOBAL accesses overlap.
#define KMAHIP_GLOBAL __attribute__((address_space(1)))
typedef KMAHIP_GLOBAL CAnk GAnk;
typedef KMAHIP_GLOBAL CSeg GSeg;
typedef KMAHIP_GLOBAL int GInt;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
static_assert(sizeof(CAnk) == 32, "an anchor is loaded as two 16-byte words");
// an anchor into registers: two 16-byte loads
__device__ __forceinline__ CAnk ank_load(const GAnk *p) {
	const KMAHIP_GLOBAL u32x4 *q = (const KMAHIP_GLOBAL u32x4 *) p;
	const u32x4 a = q[0], b = q[1];
	CAnk r;
	__builtin_memcpy(&r, &a, 16);
	__builtin_memcpy((char *) &r + 16, &b, 16);
	return r;
}

// The reference keeps three DB_size-wide arrays per thread (Score, extendScore, include; savekmers.c:5140-5160) of which a read
// touches the entries of its few candidate templates. Two homes for them:
//   DenseMap  the same arrays in the lane's HBM scratch (the lane-per-read kernel that takes every read);
//   LdsMap    a hashed table of TS entries per lane in LDS (entry (slot, lane) at slot * 64 + lane): what the fast kernel uses.
//             A template that finds the table full sets the lane's status and the read goes to the other kernel.
// A handle is obtained once per (anchor, template) and used for the three fields.
struct DenseMap {
	static constexpr bool cached_lists = false;      // (the lane-per-read kernel builds its anchors itself: their spare fields are its own)
	int *Score, *extend;
	int8_t *include;
	const int32_t *tlen;
	__device__ __forceinline__ int slot(int t, int &) const { return t; }
	__device__ __forceinline__ int TL(int h) const { return ((const KMAHIP_GLOBAL int32_t *) tlen)[h]; }
	__device__ __forceinline__ KMAHIP_GLOBAL int &S(int h) const { return ((KMAHIP_GLOBAL int *) Score)[h]; }
	__device__ __forceinline__ KMAHIP_GLOBAL int &E(int h) const { return ((KMAHIP_GLOBAL int *) extend)[h]; }
	__device__ __forceinline__ KMAHIP_GLOBAL int8_t &I(int h) const { return ((KMAHIP_GLOBAL int8_t *) include)[h]; }
};
#ifndef CHAIN_LDS_TS
#define CHAIN_LDS_TS 16
#endif
constexpr int LDS_TS = CHAIN_LDS_TS;       // slots per lane
constexpr uint32_t LDS_EMPTY = 0xFFFFFFFFu;
// (pointers into LDS carry their address space: kept in a struct as plain pointers they were compiled to FLAT accesses -- 230 M
// vector-memory instructions and not one LDS instruction per 2 M reads, SQ counters of tools/pmc_chain.sh)
#define KMAHIP_LDS __attribute__((address_space(3)))
struct LdsMap {
	static constexpr bool cached_lists = true;       // (anchors of chain_anchor_kernel carry the head of their value list)
	KMAHIP_LDS uint32_t *id;
	KMAHIP_LDS int *sc, *tl;   // (tl: the template's length, fetched once per read instead of once per anchor it occurs in)
	KMAHIP_LDS uint16_t *ex;   // (positions of reads the fast route takes fit 16 bits)
	KMAHIP_LDS int8_t *inc;
	const int32_t *tlen;
	int lane;
	__device__ __forceinline__ int slot(int t, int &status) const {
		const uint32_t h = ((uint32_t) t * 0x9E3779B1u) >> (LDS_TS == 16 ? 28 : 29);
#pragma unroll 1
		for(int x = 0; x < LDS_TS; ++x) {
			const int idx = (int) ((h + x) & (LDS_TS - 1)) * 64 + lane;
			const uint32_t cur = id[idx];
			if(cur == (uint32_t) t) return idx;
			if(cur == LDS_EMPTY) { id[idx] = (uint32_t) t; sc[idx] = 0; ex[idx] = 0; inc[idx] = 0; tl[idx] = ((const KMAHIP_GLOBAL int32_t *) tlen)[t]; return idx; }
		}
		status = 1;
		return (int) (h & (LDS_TS - 1)) * 64 + lane;          // (any slot: the read is given up)
	}
	__device__ __forceinline__ int TL(int h) const { return tl[h]; }
	__device__ __forceinline__ KMAHIP_LDS int &S(int h) const { return sc[h]; }
	__device__ __forceinline__ KMAHIP_LDS uint16_t &E(int h) const { return ex[h]; }
	__device__ __forceinline__ KMAHIP_LDS int8_t &I(int h) const { return inc[h]; }
	__device__ __forceinline__ void reset() const { for(int x = 0; x < LDS_TS; ++x) id[x * 64 + lane] = LDS_EMPTY; }
};

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
	int64_t read_base;           // number of the batch's first read in the stream the records count in (the fast route works in chunks)
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
	unsigned long long *counters;   // [0] records, [1] status, [CH_T] templates
};
constexpr int CH_T = 16;
static_assert(CH_T < KMAHIP_N_COUNTERS, "the template counter lies inside the counter block");

// what the chaining reads of the database's description, by value in the lane (read through the lane's pointer to the kernel's argument
// every field was a FLAT load of its own)
struct DbLite { uint32_t values_u16, mlen; const uint16_t *values16; const uint32_t *values32; };
template <class DB> __device__ __forceinline__ int list_n(const DB &db, uint32_t v);
template <class DB> __device__ __forceinline__ int list_at(const DB &db, uint32_t v, int i);
// length / i-th element (1-based) of an anchor's value list: from the head the anchor carries where it does (see
// chain_anchor_kernel), else from the list itself
template <class TM, class DB> __device__ __forceinline__ int ank_n(const DB &db, const CAnk &a) {
	if(TM::cached_lists) return db.values_u16 ? (int) ((uint32_t) a.score_len & 0xFFFFu) : a.score_len;
	return list_n(db, a.values);
}
template <class TM, class DB> __device__ __forceinline__ int ank_at(const DB &db, const CAnk &a, int i) {
	if(TM::cached_lists) {
		if(db.values_u16) {
			if(i == 1) return (int) ((uint32_t) a.score_len >> 16);
			if(i == 2) return (int) ((uint32_t) a.len_len & 0xFFFFu);
			if(i == 3) return (int) ((uint32_t) a.len_len >> 16);
		} else if(i == 1) return a.len_len;
	}
	return list_at(db, a.values, i);
}

template <class TM>
struct CLaneT {
	typedef TM Map;
	const DevDB *db;
	DbLite dbl;
	CAnk *VF, *VR;
	TM tm;
	int *bestT, *bestT_r;
	CSeg *tree;
	int *ovf;                    // a word of the lane's scratch: set by the tree code when it runs out of depth
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
// db_probe in two halves, for a lane that wants several probes in flight: the first bucket of a key (one round trip), and what it says --
// the k-mer's place, NOLIST, or PROBE_MORE (the bucket is full of other keys: the probe goes on in the next one, db_probe_from)
constexpr uint32_t PROBE_MORE = 0xFFFFFFFEu;
struct Bucket { u32x4 a, c; };
// (the pointers of a DevDB that a lane reads out of its CLane are generic to the compiler: FLAT loads, which count on two counters at once
// and are therefore waited for one by one -- eight buckets fetched "together" came back in eight round trips. As pointers into HBM
// (address space 1) they are GLOBAL loads and overlap.)
__device__ __forceinline__ Bucket db_bucket(const DevDB &db, uint32_t key) {
	const uint32_t b = (key * 0x9E3779B1u) >> (32u - db.nb_log2);
	const KMAHIP_GLOBAL u32x4 *p = (const KMAHIP_GLOBAL u32x4 *) (db.slots + (size_t) b * KMAHIP_BUCKET_SLOTS);
	Bucket B;
	B.a = p[0]; B.c = p[1];
	return B;
}
__device__ __forceinline__ uint32_t db_list_id(const DevDB &db, uint32_t gp) { return ((const KMAHIP_GLOBAL uint32_t *) db.vs_id)[gp]; }
__device__ __forceinline__ uint32_t db_bucket_says(const Bucket &B, uint32_t key) {
	if(B.a.x == key && B.a.y != KMAHIP_EMPTY_VI) return B.a.y;
	if(B.a.z == key && B.a.w != KMAHIP_EMPTY_VI) return B.a.w;
	if(B.c.x == key && B.c.y != KMAHIP_EMPTY_VI) return B.c.y;
	if(B.c.z == key && B.c.w != KMAHIP_EMPTY_VI) return B.c.w;
	return B.c.w == KMAHIP_EMPTY_VI ? NOLIST : PROBE_MORE;
}
// (a probe whose first bucket said PROBE_MORE: from the bucket behind it on)
__device__ __forceinline__ uint32_t db_probe_from(const DevDB &db, uint32_t key) {
	const uint32_t sh = 32u - db.nb_log2, nbm = (1u << db.nb_log2) - 1u;
	uint32_t b = (((key * 0x9E3779B1u) >> sh) + 1u) & nbm;
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
template <class DB> __device__ __forceinline__ int list_n(const DB &db, uint32_t v) { return db.values_u16 ? (int) ((const KMAHIP_GLOBAL uint16_t *) db.values16)[v] : (int) ((const KMAHIP_GLOBAL uint32_t *) db.values32)[v]; }
template <class DB> __device__ __forceinline__ int list_at(const DB &db, uint32_t v, int i) { return db.values_u16 ? (int) ((const KMAHIP_GLOBAL uint16_t *) db.values16)[v + i] : (int) ((const KMAHIP_GLOBAL uint32_t *) db.values32)[v + i]; }

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
	// (the tree and the overflow word by pointer, not the lane: a lane whose address goes into a function that is not inlined lives in
	// scratch as a whole -- every counter, every table pointer a memory access)
	__device__ __noinline__ static unsigned add(GSeg *v, GInt *ovf, int root, int node) {
		GSeg &R = v[root], &Nn = v[node];
		if(R.b0 >= 0) {
			if(Nn.start < R.start && R.end < Nn.end) {
				R.start = Nn.start; R.end = Nn.end; R.covered = Nn.covered; Nn.covered = 0; R.b0 = -1;
				return R.covered;
			} else if(R.end < Nn.end) R.end = Nn.end;
			else if(Nn.start < R.start) R.start = Nn.start;
			unsigned pos = v[R.b1].start;
			if(Nn.end < pos) R.covered = v[R.b1].covered + SegOps<D - 1>::add(v, ovf, R.b0, node);
			else if(pos <= Nn.start) R.covered = v[R.b0].covered + SegOps<D - 1>::add(v, ovf, R.b1, node);
			else {
				pos = Nn.start;
				Nn.start = v[R.b0].end + 1;
				Nn.covered = Nn.end - Nn.start;
				const unsigned covered = SegOps<D - 1>::add(v, ovf, R.b1, node);
				Nn.start = pos;
				Nn.end = v[R.b0].end;
				Nn.covered = Nn.end - Nn.start;
				R.covered = covered + SegOps<D - 1>::add(v, ovf, R.b0, node);
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
	__device__ __noinline__ static unsigned que(const GSeg *v, int i, unsigned start, unsigned end) {
		const GSeg &s = v[i];
		if(end < s.start || s.end < start) return 0;
		if(start <= s.start && s.end <= end) return s.covered;
		if(s.b0 >= 0) return SegOps<D - 1>::que(v, s.b0, start, end) + SegOps<D - 1>::que(v, s.b1, start, end);
		if(s.start <= start && end <= s.end) return end - start;
		if(s.start <= start && start < s.end) return s.end - start;
		if(s.start < end && end <= s.end) return end - s.start;
		return 0;
	}
};
template <> struct SegOps<0> {
	__device__ static unsigned add(GSeg *, GInt *ovf, int, int) { *ovf = 1; return 0; }          // (deeper than SEG_DEPTH: the read is given up)
	__device__ static unsigned que(const GSeg *, int, unsigned, unsigned) { return 0; }
};
constexpr int SEG_DEPTH = 24;

template <class CLane> __device__ void seg_grow(CLane &L, unsigned start, unsigned end) {
	if(L.s_cap <= L.tree_n + 2) { L.status = 1; return; }
	GSeg *v = (GSeg *) L.tree;
	if(L.tree_n == 0) {
		L.tree_n = 1;
		v[0].start = start; v[0].end = end; v[0].covered = end - start; v[0].b0 = v[0].b1 = -1;
		return;
	}
	const int node = L.tree_n;
	v[node].start = start; v[node].end = end; v[node].covered = end - start; v[node].b0 = -1;
	GInt *ovf = (GInt *) L.ovf;
	v[0].covered = SegOps<SEG_DEPTH>::add(v, ovf, 0, node);
	if(*ovf) { L.status = 1; *ovf = 0; }
	if(v[node].covered) L.tree_n += 2;
}

// ---- chaining helpers -------------------------------------------------------------------------------------------------------
template <class CLane> __device__ __forceinline__ int bridge(const CLane &L, int mlen, int weight, int gaps) {
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
// *cstart (may be NULL): where that anchor starts (the caller's V[prev].start, without the load).
template <class CLane> __device__ int chain_templates(CLane &L, GAnk *V, int src, GInt *bests, int room, int *cstart = nullptr) {
	const DbLite db = L.dbl;
	if(src < 0) return -1;
	int nextAnker = 0;
	typedef typename CLane::Map Map;
	// (the source anchor once: it is the first anchor of the walk and holds the chain's score -- three dependent loads were one)
	const CAnk a = ank_load(&V[src]);
	{
		const int n = ank_n<Map>(db, a);
		if(n + 1 > room) { L.status = 1; bests[0] = 0; return -1; }
		bests[0] = n;
		for(int i = n; i >= 1; --i) {
			const int t = ank_at<Map>(db, a, i);
			bests[i] = t;
			if(++L.tm.I(L.tm.slot(t, L.status)) == 1) nextAnker = 1;
		}
	}
	const int bestScore = a.score;
	int prev = src, prev_start = (int) a.start;
	// (the anchor in hand in registers, the one below it and the head of its list asked for ahead: see the chaining loop)
	CAnk nxt = a;
	int nxt_n = nextAnker ? ank_n<Map>(db, nxt) : 0;
	for(int node = src; nextAnker && node >= 0; --node) {
		const CAnk cur = nxt;
		const int n = nxt_n;
		if(node > 0) { nxt = ank_load(&V[node - 1]); nxt_n = ank_n<Map>(db, nxt); }
		const int start = (int) cur.start, end = (int) cur.end;
		bool silenced = false;
		for(int i = n; i >= 1; --i) {
			const int t = ank_at<Map>(db, cur, i);
			const int th = L.tm.slot(t, L.status);
			if(!L.tm.I(th)) continue;
			int score = L.tm.S(th);
			const int pos = L.tm.E(th);
			if(pos == 0) score = cur.weight;
			else {
				score += bridge(L, (int) db.mlen, cur.weight, pos - end);
				silenced = true;
			}
			if(bestScore <= score) {
				int tmp;
				if(start) {
					tmp = L.W1 + (start - 1) * L.U;
					tmp = score + (L.Wl < tmp ? tmp : L.Wl);
				} else tmp = score;
				if(tmp == bestScore) { score = bestScore; nextAnker = 0; prev = node; prev_start = start; }
			}
			L.tm.E(th) = start;
			L.tm.S(th) = score;
		}
		if(silenced) V[node].score = 0;
	}
	int j = 0;
	for(int i = 1; i <= bests[0]; ++i) {
		const int t = bests[i];
		const int th = L.tm.slot(t, L.status);
		if(L.tm.I(th) == 1 && bestScore <= L.tm.S(th)) bests[++j] = t;
		L.tm.S(th) = 0; L.tm.I(th) = 0; L.tm.E(th) = 0;
	}
	bests[0] = j;
	if(cstart) *cstart = prev_start;
	return j ? prev : -1;
}

__device__ int prune(GAnk *V, int head, int k) {
	while(head >= 0 && V[head].score < k) head = V[head].descend;
	if(head < 0) return -1;
	int prev = head;
	for(int node = V[head].descend; node >= 0; node = V[node].descend) if(k <= V[node].score) { V[prev].descend = node; prev = node; }
	V[prev].descend = -1;
	return head;
}

__device__ int best_anker(GAnk *V, int *head, unsigned *ties) {
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

__device__ int tie_anker(const GAnk *V, int stop, int src, int best) {
	if(src < 0 || (int) V[src].start <= stop) return -1;
	while(src > 0 && stop < (int) V[--src].start) if(V[src].score == V[best].score) return src;
	return -1;
}

struct ChEnd { int score; unsigned end; };      // what choose_chain reads of a strand's best anchor
__device__ int choose_chain(const ChEnd &b, const ChEnd &r, int cStart, int cStart_r, double coverT, int *Start, int *Len) {
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

#ifndef CHAIN_PROBE_BLOCK
#define CHAIN_PROBE_BLOCK 8
#endif
constexpr int PB = CHAIN_PROBE_BLOCK;         // probes a lane of the slow route keeps in flight (build_ankers)
static_assert(PB >= 1 && PB + 15 <= 32, "a block's k-mers come out of one 32-base window");
// anchors of one strand in forward coordinates (savekmers.c:5208-5330, 5333-5452); returns their number
// (a function of its own, not inlined into the kernel: there its loop shared 254 registers and 2 kB of scratch per lane with the chaining
// and the extraction, and kept its counters in scratch -- a round trip per position; the views and the database come in by value for the
// same reason)
template <class CLane> __attribute__((noinline)) __device__ int build_ankers(CLane &L, const QView qf, const QView qr, int exhaustive, int is_rc, CAnk *V_) {
	GAnk *const V = (GAnk *) V_;
	const DevDB &db = *L.db;
	const int k = L.k, seqlen = qf.L, nN = qf.nN;
	const int M_ = L.M, MM_ = L.MM, a_cap_ = L.a_cap;
	// (what the fetches of a block need of the database, once: pointers into HBM)
	const KMAHIP_GLOBAL u32x4 *const g_slots = (const KMAHIP_GLOBAL u32x4 *) db.slots;
	const KMAHIP_GLOBAL uint32_t *const g_vs_id = (const KMAHIP_GLOBAL uint32_t *) db.vs_id;
	const uint32_t g_sh = 32u - db.nb_log2;
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
		// A lane walks its read alone: with a probe (or a step along the template store) per position, every position of every lane is a
		// round trip to memory the whole wavefront waits for -- 20 000 of them for a 10 kb read. So the lists of the next PB positions are
		// fetched together: one round trip for the read's window, one for the PB buckets, one for the lists of the k-mers that are there;
		// the state machine below then runs on registers. (PB = 1: a probe or a step per position, as first written.)
		uint32_t pre[PB];
		int pre_at = 0, pre_n = 0;             // the block's answers: pre[pre_at .. pre_n) are still to come
		for(; j < segend - k + 1; ++j, --rcpos) {
			if(PB > 1 && pre_at == pre_n) {
				pre_n = min(PB, segend - k + 1 - j); pre_at = 0;
				// the block's k-mers out of ONE 32-base window of the read (PB + k - 1 <= 32 bases): a round trip for the window, one for
				// the PB buckets. The reverse strand's windows move backwards: position rcpos - x lies pre_n - 1 - x bases into the
				// window that starts at the block's last position; a block that hangs over an end of the read (the reference's buffer
				// has zeros there, rc_kmer) takes its k-mers one by one.
				uint32_t keys[PB];
				const int p0 = rcpos - (pre_n - 1);
				if(!is_rc || (p0 >= 0 && rcpos + k <= seqlen)) {
					const uint64_t w = is_rc ? qwin(qr, p0) : qwin(qf, j);
#pragma unroll
					for(int x = 0; x < PB; ++x) {
						const int y = is_rc ? max(pre_n - 1 - x, 0) : x;
						keys[x] = (uint32_t) ((w << (2 * y)) >> (64 - 2 * k));
					}
				} else {
#pragma unroll
					for(int x = 0; x < PB; ++x) keys[x] = x < pre_n ? rc_kmer(qr, rcpos - x, k) : 0u;
				}
				// (a key whose bucket is full of other keys lives further on: the unresolved ones of the block go on together, a round trip
				// per round for the block instead of one per position -- with 64 lanes some lane has such a key at nearly every position)
				uint32_t bi[PB];
				const uint32_t g_nbm = (1u << db.nb_log2) - 1u;
#pragma unroll
				for(int x = 0; x < PB; ++x) bi[x] = (keys[x] * 0x9E3779B1u) >> g_sh;
				for(int round = 0; round < 32; ++round) {
					Bucket bk[PB];
#pragma unroll
					for(int x = 0; x < PB; ++x) {          // (every lane loads for every position, beyond pre_n and where the answer is there too: no branches)
						const KMAHIP_GLOBAL u32x4 *pb = g_slots + (size_t) bi[x] * (KMAHIP_BUCKET_SLOTS / 2);
						bk[x].a = pb[0]; bk[x].c = pb[1];
					}
					bool more = false;
#pragma unroll
					for(int x = 0; x < PB; ++x) {
						const uint32_t says = db_bucket_says(bk[x], keys[x]);
						pre[x] = (round == 0 || pre[x] == PROBE_MORE) ? says : pre[x];
						more = more || (x < pre_n && pre[x] == PROBE_MORE);
						bi[x] = (bi[x] + 1u) & g_nbm;
					}
					if(!more) break;
				}
				// ... and a third for the value lists of the k-mers that are there: pre[] holds a position's list (NOLIST: none) from here on
				// (every lane loads for every position -- entry 0 where there is nothing to look up --, so that the loads are not eight branches
				// with a wait behind each)
				uint32_t lid[PB];
#pragma unroll
				for(int x = 0; x < PB; ++x) lid[x] = g_vs_id[pre[x] < PROBE_MORE ? pre[x] : 0u];
#pragma unroll
				for(int x = 0; x < PB; ++x) pre[x] = pre[x] < PROBE_MORE ? lid[x] : pre[x];
			}
			uint32_t fetched = NOLIST;
			if(PB > 1) {
#pragma unroll
				for(int x = 0; x < PB; ++x) if(x == pre_at) fetched = pre[x];
				++pre_at;
			}
			// A read that matches a template keeps matching it: when the base that enters the window is the template's next base
			// (the one in front of it on the reverse strand, whose windows move backwards), the k-mer is the template's
			// neighbouring k-mer and its value list stands in vs_id -- no probe (the walk of scan.hip, one lane here)
			uint32_t values;
			bool walked = false;
			if(PB > 1) {
				// (no walk here: what it saves is the probe, and the probes are there already; the list of a k-mer is the same either way)
				values = fetched;
				if(values == PROBE_MORE) {
					const uint32_t g2 = db_probe_from(db, is_rc ? rc_kmer(qr, rcpos, k) : q_kmer(qf, j, k));
					values = g2 == NOLIST ? NOLIST : db.vs_id[g2];
				}
				walked = true;
			} else if(gp != NOLIST) {
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
						V[v].weight = Ms * M_ + MMs * MM_;
						V[v].end = (unsigned) (j - gaps + k);
						V[v].descend = v + 1;
						++v;
						if(v >= a_cap_) { L.status = 1; return 0; }
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
		V[v].weight = Ms * M_ + MMs * MM_;
		V[v].end = (unsigned) (seqlen - gaps);
	}
	return hits;
}

struct Emit {
	const ChainArgs *A;
	int64_t read;
	int ordinal;
};

// One S2 record. The lanes of a wavefront that emit at the same moment take their slots with ONE atomic per counter: two atomics
// per record on two addresses, as first written, ran at 17 ns a record -- 35 of the 51 ms the kernel took for 2 M reads.
template <class CLane> __device__ void emit_record(CLane &L, Emit &E, int rc_flag, int emit_rc, int q_start, int q_end, const GInt *bt) {
	const ChainArgs &A = *E.A;
	const int nT = bt[0];
	const int lane = (int) (threadIdx.x & 63);
	const unsigned long long act = __ballot(1);                       // the lanes that are here together
	const unsigned long long below = act & ((1ull << lane) - 1ull);
	const int leader = __ffsll((long long) act) - 1;
	// exclusive prefix and total of nT over those lanes, one ballot per bit that any of them has set
	unsigned pre = 0, tot = 0;
	for(int b = 0; b < 31 && __ballot((nT >> b) != 0); ++b) {
		const unsigned long long m = __ballot((nT >> b) & 1);
		pre += (unsigned) __popcll(m & below) << b;
		tot += (unsigned) __popcll(m) << b;
	}
	unsigned long long slot0 = 0, toff0 = 0;
	if(lane == leader) {
		slot0 = atomicAdd(&A.counters[0], (unsigned long long) __popcll(act));
		toff0 = atomicAdd(&A.counters[CH_T], (unsigned long long) tot);      // (a 128-byte line away from the record counter: atomics on one line are served in turn)
	}
	// (the first active lane is the leader: a scalar broadcast, no lane index in the instruction stream)
	const unsigned lo0 = (unsigned) __builtin_amdgcn_readfirstlane((int) (slot0 & 0xFFFFFFFFull)), hi0 = (unsigned) __builtin_amdgcn_readfirstlane((int) (slot0 >> 32));
	const unsigned lo2 = (unsigned) __builtin_amdgcn_readfirstlane((int) (toff0 & 0xFFFFFFFFull)), hi2 = (unsigned) __builtin_amdgcn_readfirstlane((int) (toff0 >> 32));
	const unsigned long long slot = (((unsigned long long) hi0 << 32) | lo0) + (unsigned long long) __popcll(below);
	const unsigned long long toff = (((unsigned long long) hi2 << 32) | lo2) + pre;
	if((int64_t) slot >= A.rec_cap || (int64_t) (toff + nT) > A.T_cap) { atomicMax(&A.counters[1], 2ull); ++E.ordinal; return; }
	KMAHIP_GLOBAL int32_t *r = (KMAHIP_GLOBAL int32_t *) A.rec + 8 * slot;
	r[0] = (int32_t) (E.read & 0xFFFFFFFFll); r[1] = (int32_t) (E.read >> 32); r[2] = E.ordinal++; r[3] = rc_flag; r[4] = emit_rc;
	r[5] = q_start; r[6] = q_end; r[7] = nT;
	((KMAHIP_GLOBAL int64_t *) A.rec_T)[slot] = (int64_t) toff;
	KMAHIP_GLOBAL int32_t *To = (KMAHIP_GLOBAL int32_t *) A.T;
	for(int i = 0; i < nT; ++i) To[toff + i] = bt[1 + i];
}

// the read's anchors are in L.VF / L.VR (hitF / hitR of them; V[0] holds start 0, end 0, descend -1 when a strand has none):
// chaining and extraction, savekmers.c:5466-5940
template <class CLane> __device__ void chain_read_tail(CLane &L, const ChainArgs &A, int64_t r, int seqlen, unsigned hitF, unsigned hitR);

template <class CLane> __device__ void chain_read(CLane &L, const ChainArgs &A, int64_t r) {
	const int k = L.k;
	QView qf;
	qf.w = A.seq + A.seq_off[r]; qf.L = A.len[r]; qf.N = A.N + A.N_off[r]; qf.nN = (int) (A.N_off[r + 1] - A.N_off[r]); qf.rc = 0;
	QView qr = qf; qr.rc = 1;
	const int seqlen = qf.L;
	if(seqlen < k) return;
	if(seqlen + 2 > L.a_cap) { L.status = 1; return; }
	const unsigned hitF = (unsigned) build_ankers(L, qf, qr, A.exhaustive, 0, L.VF);
	const unsigned hitR = (unsigned) build_ankers(L, qf, qr, A.exhaustive, 1, L.VR);
	if(L.status || (!hitF && !hitR)) return;
	if(A.stop_after == 1) return;
	chain_read_tail(L, A, r, seqlen, hitF, hitR);
}

template <class CLane> __device__ void chain_read_tail(CLane &L, const ChainArgs &A, int64_t r, int seqlen, unsigned hitF, unsigned hitR) {
	const DbLite db = L.dbl;
	const int k = L.k;
	GAnk *VF = (GAnk *) L.VF, *VR = (GAnk *) L.VR;
	GInt *bestT = (GInt *) L.bestT, *bestT_r = (GInt *) L.bestT_r;
	L.tree_n = 0;
	Emit E = {&A, A.read_base + r, 0};

	// chains left to right, per strand (savekmers.c:5466-5634)
	GAnk *best = nullptr, *best_r = &VF[0];
	unsigned ties = 0;
	int iF = 0, sF = 0, eF = 0, iR = 0, sR = 0, eR = 0;      // index, score and end of the forward / reverse strand's best anchor
	int a_min = 0x7fffffff, a_max = 0;          // every anchor of either strand lies inside [a_min, a_max)
	VF[0].score = 0;
	{
		GInt *bests = bestT;
		bestT[0] = 0; bestT_r[0] = 0;
		for(int strand = 0; strand < 2; ++strand) {
			GAnk *V = strand ? VR : VF;
			unsigned HIT = (strand ? hitR : hitF) + 1;
			if(strand) {
				V[0].score = 0;          // (score_len / len_len of the anchor in hand live in registers below)
				bests = bestT_r;
				best = best_r;
				best_r = &V[0];
			}
			bests[0] = 0;
			int vi = 0;
			// The anchor in hand is kept in registers (its fields would be re-read from HBM behind every store otherwise), and so is
			// what the comparisons below read of the best anchor so far; best_r points at V[0] BEFORE that anchor is chained, so the
			// first comparison of a strand compares the anchor with itself -- an equality, which counts a tie (kept, b_idx == vi).
			// The next anchor and the head of its value list are asked for while this one is worked on.
			int b_idx = 0, b_score = 0, b_sl = 0, b_end = 0, n_bests = 0;
			typedef typename CLane::Map Map;
			CAnk nxt = ank_load(&V[0]);
			int nxt_n = HIT > 1 ? ank_n<Map>(db, nxt) : 0;
			while(--HIT) {
				const CAnk cur = nxt;
				const int n = nxt_n;
				if(HIT > 1) { nxt = ank_load(&V[vi + 1]); nxt_n = ank_n<Map>(db, nxt); }
				const int start = (int) cur.start, end = (int) cur.end, weight = cur.weight;
				a_min = start < a_min ? start : a_min; a_max = end > a_max ? end : a_max;
				int a_score = 0, a_sl = 0, a_ll = 1;
				for(int i = n; i >= 1; --i) {
					const int t = ank_at<Map>(db, cur, i);
					const int th = L.tm.slot(t, L.status);
					if(L.status) return;
					int score = L.tm.S(th);
					const int pos = L.tm.E(th);
					if(!L.tm.I(th)) {
						L.tm.I(th) = 1;
						if(n_bests + 2 > L.b_cap) { L.status = 1; return; }
						bests[++n_bests] = t;
						if(start) {
							score = L.W1 + (start - 1) * L.U;
							score = weight + (L.Wl < score ? score : L.Wl);
						} else score = weight;
					} else {
						score += bridge(L, (int) db.mlen, weight, start - pos);
						if(score < 0) {
							int test = start ? L.W1 + (start - 1) * L.U : 0;
							if(test < L.Wl) test = L.Wl;
							if(score < test + weight) score = test + weight;
						}
					}
					if(a_score < score) a_score = score;
					int len_len = L.tm.TL(th);
					if(seqlen < len_len) len_len = seqlen;
					double score_len = score;
					if(a_ll != len_len) { score_len /= len_len; score_len *= a_ll; }
					if(a_sl < score_len || (a_sl == score_len && a_sl < score)) { a_sl = score; a_ll = len_len; }
					L.tm.S(th) = score;
					L.tm.E(th) = end;
				}
				V[vi].score = a_score;
				const int c_score = b_idx == vi ? a_score : b_score, c_sl = b_idx == vi ? a_sl : b_sl;
				if(c_score < a_score) { best_r = &V[vi]; ties = 0; }
				else if(c_score == a_score) {
					if(c_sl < a_sl) { best_r = &V[vi]; ties = 0; }
					else { best_r = &V[vi]; ++ties; }
				}
				if(best_r == &V[vi]) { b_idx = vi; b_score = a_score; b_sl = a_sl; b_end = end; }
				++vi;
			}
			// (what the extraction asks of the strand's best anchor, kept in registers: each was a dependent load from HBM)
			if(strand) { iR = b_idx; sR = b_score; eR = b_end; } else { iF = b_idx; sF = b_score; eF = b_end; }
			bests[0] = n_bests;
			for(int i = 1; i <= bests[0]; ++i) { const int th = L.tm.slot(bests[i], L.status); L.tm.S(th) = 0; L.tm.E(th) = 0; L.tm.I(th) = 0; }
		}
	}
	if(sF < k && sR < k) return;
	if(A.stop_after == 2) return;

	const int VF_start = (int) VF[0].start, VR_start = (int) VR[0].start;
	// pruneAnkers (kmeranker.c:372-398) drops the anchors below k from the strands' lists, which only the search for the NEXT chain
	// walks (best_anker): it is put off until a read gets that far -- the usual read ends after its first chain, and the walks are
	// dependent loads from HBM, an anchor at a time. What the first chain needs of it is whether a strand has an anchor of k or more at
	// all: its best one has. (Anchors silenced or taken in between have score 0: the search drops those from the lists by itself.)
	int headF = 0, headR = 0;
	bool pruned = false;
	if(sF < k) { sF = 0; best->score = 0; }
	if(sR < k) { sR = 0; best_r->score = 0; }
	bestT[0] = 0; bestT_r[0] = 0;
	int bi = iF, bri = iR;
	int cStart = -1, cStart_r = -1, start = 0, len = 0, rc;
	if(!sF || !sR) {
		if(sF) {
			if(chain_templates(L, VF, bi, bestT, L.b_cap, &cStart) < 0) return;
			start = cStart; len = eF - start; rc = 1;
		} else {
			if(chain_templates(L, VR, bri, bestT_r, L.b_cap, &cStart_r) < 0) return;
			start = cStart_r; len = eR - start; rc = 2;
		}
	} else {
		if(chain_templates(L, VF, bi, bestT, L.b_cap, &cStart) < 0) return;
		if(chain_templates(L, VR, bri, bestT_r, L.b_cap, &cStart_r) < 0) return;
		const ChEnd cb = {sF, (unsigned) eF}, cr = {sR, (unsigned) eR};
		rc = choose_chain(cb, cr, cStart, cStart_r, A.coverT, &start, &len);
	}
	if(len < A.minlen || (sF > sR ? sF : sR) < k) return;
	if(A.stop_after == 3) return;
	while((bi >= 0 || bri >= 0) && !L.status) {
		if(ties) {
			for(int side = 0; side < 2; ++side) {
				if(!(rc & (1 << side))) continue;
				GAnk *V = side ? VR : VF;
				GInt *bt = side ? bestT_r : bestT;
				const int bidx = side ? bri : bi, vstart = side ? VR_start : VF_start;
				int v = bidx;
				while((v = tie_anker(V, start < vstart ? vstart : start, v, bidx)) >= 0) {
					if((double) (unsigned) (V[v].end - (unsigned) start) < A.coverT * len) break;      // (unsigned arithmetic in the reference)
					for(int i = 1; i <= bt[0]; ++i) { const int th = L.tm.slot(bt[i], L.status); L.tm.I(th) = 1; L.tm.S(th) = 0; L.tm.E(th) = 0; }
					GInt *tail = bt + bt[0];
					const int keep = *tail;
					*tail = 0;
					chain_templates(L, V, v, tail, L.b_cap - bt[0]);
					bt[0] += *tail;
					*tail = keep;
					if(L.status) return;
				}
				for(int i = 1; i <= bt[0]; ++i) { const int th = L.tm.slot(bt[i], L.status); L.tm.I(th) = 0; L.tm.S(th) = 0; L.tm.E(th) = 0; }
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
			// Nothing more can come once the one stretch [S, E) taken so far holds (nearly) every anchor. A later chain [cs, end) is
			// cut out of [a_min, a_max); with S <= a_min and d = a_max - E bases beyond the stretch (an anchor closed by the next
			// one ends k + 1 behind its last hit, the read's last anchor at its last hit, so d is a few bases):
			//   cs < E: the overlap is min(end, E) - cs >= 1 of at most (E - cs) + d bases; it is turned down when
			//           overlap > coverT * length, which holds for every cs once 1 - coverT > coverT * d;
			//   cs >= E: the chain is at most d bases long, below minlen.
			// The reference goes through every remaining anchor, walks its chain back and turns it down (savekmers.c:5860-5925);
			// the usual read ends here.
			if(A.stop_after == 4) return;
			if(L.tree_n == 1 && (int) ((const GSeg *) L.tree)[0].start <= a_min) {
				const int d = a_max - (int) ((const GSeg *) L.tree)[0].end;
				if(d <= 0 ? (A.coverT < 1.0 && A.minlen > 0) : (d < A.minlen && 1.0 - A.coverT > A.coverT * d + 1e-9)) break;
			}
		}
		// next chain of either strand (savekmers.c:5827-5925)
		if(!pruned) { headF = prune(VF, 0, k); headR = prune(VR, 0, k); pruned = true; }
		ties = 0;
		rc = 0;
		for(int side = 0; side < 2; ++side) {
			GAnk *V = side ? VR : VF;
			GInt *bt = side ? bestT_r : bestT;
			int &bidx = side ? bri : bi, &head = side ? headR : headF, &cs = side ? cStart_r : cStart;
			if(bidx < 0) continue;
			bool ok = false;
			if(V[bidx].score) {
				const int s = chain_templates(L, V, bidx, bt, L.b_cap);
				if(s >= 0) {
					cs = (int) V[s].start;
					const int cover = (int) SegOps<SEG_DEPTH>::que((const GSeg *) L.tree, 0, (unsigned) cs, V[bidx].end);
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
							const int cover = (int) SegOps<SEG_DEPTH>::que((const GSeg *) L.tree, 0, (unsigned) cs, V[bidx].end);
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
		if(bi >= 0 && bri >= 0) {
			const CAnk ab = ank_load(&VF[bi]), ar = ank_load(&VR[bri]);
			const ChEnd cb = {ab.score, ab.end}, cr = {ar.score, ar.end};
			rc = choose_chain(cb, cr, cStart, cStart_r, A.coverT, &start, &len);
		}
		else if(bi >= 0) { rc = 1; start = cStart; len = (int) VF[bi].end - start; }
		else { rc = 2; start = cStart_r; len = (int) VR[bri].end - start; }
	}
}

#ifndef CHAIN_MIN_WAVES
#define CHAIN_MIN_WAVES 2
#endif
// The lane-per-read kernel: every read it is given (list == NULL: all of them), anchors included, on per-lane scratch in HBM.
// Takes what the fast pair of kernels below leaves: reads with N's, long reads, reads with more anchors or candidate templates
// than the fast kernels' LDS tables hold.
__global__ __launch_bounds__(64, CHAIN_MIN_WAVES) void chain_kernel(const ChainArgs A, const int64_t *list, int64_t n_list) {
	const int64_t lane = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(lane >= A.lanes) return;
	uint8_t *base = A.scratch + lane * A.lane_bytes;
	const int64_t D = A.db.DB_size;
	CLaneT<DenseMap> L;
	L.db = &A.db;
	L.dbl.values_u16 = A.db.values_u16; L.dbl.mlen = A.db.mlen; L.dbl.values16 = A.db.values16; L.dbl.values32 = A.db.values32;
	L.VF = (CAnk *) base; base += (size_t) A.a_cap * sizeof(CAnk);
	L.VR = (CAnk *) base; base += (size_t) A.a_cap * sizeof(CAnk);
	L.tm.Score = (int *) base; base += (size_t) (D + 1) * 4;
	L.tm.extend = (int *) base; base += (size_t) (D + 1) * 4;
	L.bestT = (int *) base; base += (size_t) A.b_cap * 4;
	L.bestT_r = (int *) base; base += (size_t) A.b_cap * 4;
	L.tree = (CSeg *) base; base += (size_t) A.s_cap * sizeof(CSeg);
	L.ovf = (int *) (L.tree + (A.s_cap - 1));          // (the last node is never used: seg_grow stops two short of s_cap)
	*(GInt *) L.ovf = 0;
	L.tm.include = (int8_t *) base;
	L.tm.tlen = A.db.tlen;
	L.k = (int) A.db.kmersize; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1; L.Wl = A.Wl;
	L.a_cap = A.a_cap; L.b_cap = A.b_cap; L.s_cap = A.s_cap;
	L.status = 0; L.tree_n = 0;
	const int64_t total = list ? n_list : A.n_reads;
	for(int64_t x = lane; x < total; x += A.lanes) {
		chain_read(L, A, list ? list[x] : x);
		if(L.status) {
			// leave the per-template arrays clean for the next read of this lane, and say so
			atomicMax(&A.counters[1], 40ull);
			for(int64_t t = 0; t <= D; ++t) { L.tm.Score[t] = 0; L.tm.extend[t] = 0; L.tm.include[t] = 0; }
			L.status = 0;
		}
	}
}

// ---- long reads: the anchors by a wavefront per read and strand, chaining and extraction by a lane per read on ready anchors -----------
// A lane that walks a 10 kb read alone makes 20 000 lookups one block after the other, and the wavefront lasts as long as its longest
// read (build_ankers above: 28 ms for 2 000 reads however few they are). The lookups of a read are independent of each other; what is
// serial is the anchor state machine -- and that is a scan: whether a hit opens an anchor or continues one depends on the hit before it
// only (the same list, 0 or exactly k starts missed in between, savekmers.c:5262-5300). So a wavefront takes a strand LA_P k-mer starts
// at a time, LA_SEG to a lane: every lane's buckets travel together, then the lists of the k-mers that are there; the hit before a lane's
// segment comes from a "last non-empty" scan over the wavefront, the anchor numbers from a prefix sum of the opens, weights and last
// hits are summed per anchor in LDS (an anchor may span lanes and passes: slot 0 holds the one that is still open when a pass ends).
// The anchors of a read lie in a region of the pool sized by its k-mer starts (v_off: no counting pass, no atomics), the forward
// strand's first; chain_long_tail_kernel then runs chain_read_tail on them as chain_kernel does on the ones it built itself.
// Reads with N's stay with chain_kernel (the reference restarts the reverse strand k bases off behind an N, savekmers.c:5447-5449).
constexpr int LA_SEG = 8, LA_P = 64 * LA_SEG;
static_assert(LA_SEG + 15 <= 32, "a lane's k-mers come out of one 32-base window");
struct LongArgs {
	const int64_t *list;      // the reads (of ChainArgs' batch) this route takes
	int64_t first, count;     // list[first .. first + count) in this launch
	const int64_t *v_off;     // per list entry: first anchor of its region in the pool, counted from v_off[first] (2 * (k-mer starts + 1) anchors each)
	CAnk *pool;
	int32_t *hits;            // 2 per list entry: anchors of the forward / the reverse strand
};
__device__ __forceinline__ void ank_store(GAnk *p, const CAnk &a) {
	u32x4 lo, hi;
	__builtin_memcpy(&lo, &a, 16);
	__builtin_memcpy(&hi, (const char *) &a + 16, 16);
	KMAHIP_GLOBAL u32x4 *q = (KMAHIP_GLOBAL u32x4 *) p;
	q[0] = lo; q[1] = hi;
}
__global__ __launch_bounds__(256) void chain_long_sizes_kernel(int64_t n_list, const int64_t *list, const int32_t *len, int k, int64_t *cnt) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x < n_list) cnt[x] = 2 * ((int64_t) max(len[list[x]] - k + 1, 0) + 1);
}
__global__ __launch_bounds__(64) void chain_long_anchor_kernel(const ChainArgs A, const LongArgs G) {
	__shared__ int s_w[LA_P + 1];
	__shared__ uint32_t s_last[LA_P + 1], s_start[LA_P + 1], s_val[LA_P + 1];
	const int lane = (int) threadIdx.x;
	const int64_t x = G.first + (int64_t) (blockIdx.x >> 1);
	const int strand = (int) (blockIdx.x & 1u);
	const int64_t r = G.list[x];
	const int k = (int) A.db.kmersize, L = A.len[r], npos = L - k + 1;
	GAnk *const V = (GAnk *) (G.pool + (G.v_off[x] - G.v_off[G.first]) + (strand ? (int64_t) max(npos, 0) + 1 : 0));
	QView q;
	q.w = A.seq + A.seq_off[r]; q.L = L; q.N = nullptr; q.nN = 0; q.rc = strand;
	const KMAHIP_GLOBAL u32x4 *const g_slots = (const KMAHIP_GLOBAL u32x4 *) A.db.slots;
	const KMAHIP_GLOBAL uint32_t *const g_vs_id = (const KMAHIP_GLOBAL uint32_t *) A.db.vs_id;
	const KMAHIP_GLOBAL uint32_t *const g_kbits = (const KMAHIP_GLOBAL uint32_t *) A.db.kbits;
	const uint32_t g_sh = 32u - A.db.nb_log2, kb_sh = A.db.kbits_shift;
	const int kM = k * A.M, M1 = A.M, kMM = k * A.M + A.MM;
	CAnk none;
	none.score = 0; none.weight = 0; none.score_len = 0; none.len_len = 0; none.start = 0; none.end = 0; none.values = NOLIST; none.descend = -1;
	// the reference looks at every k-th k-mer of a strand first and leaves the strand alone when none of them is known (savekmers.c:5216-5232)
	bool HIT = A.exhaustive != 0;
	if(!HIT && npos > 0) {
		bool any = false;
		for(int j = lane * k; j < npos; j += 64 * k) any = any || db_probe(A.db, (uint32_t) (qwin(q, j) >> (64 - 2 * k))) != NOLIST;
		HIT = __any(any);
	}
	if(!HIT || npos <= 0) {
		if(lane == 0) { ank_store(&V[0], none); G.hits[2 * x + strand] = 0; }
		return;
	}
	for(int s = lane; s <= LA_P; s += 64) { s_w[s] = 0; s_last[s] = 0; }
	__syncthreads();
	int carry_h = -1, carry_cnt = 0;
	uint32_t carry_v = NOLIST;
	for(int c0 = 0; c0 < npos; c0 += LA_P) {
		// the lists of this lane's LA_SEG k-mer starts, forward positions j0 .. j0 + LA_SEG - 1 (on the reverse strand forward position j is
		// position npos - 1 - j of the reverse complement: the lane's window starts at its LAST position there)
		const int j0 = c0 + lane * LA_SEG;
		uint32_t v[LA_SEG];
		{
			uint32_t keys[LA_SEG], res[LA_SEG];
			const int p_hi = npos - 1 - j0, base = max(p_hi - (LA_SEG - 1), 0);
			const uint64_t w = j0 < npos ? qwin(q, strand ? base : j0) : 0ull;
#pragma unroll
			for(int i = 0; i < LA_SEG; ++i) {
				const int sh = strand ? max(p_hi - i - base, 0) : i;
				keys[i] = (uint32_t) ((w << (2 * sh)) >> (64 - 2 * k));
			}
			// a k-mer the presence bits do not know (small databases have them, db.hip) is not in the table: its lookup goes to bucket 0
			// with every other such lookup -- one line for all of them instead of a line each -- and its answer is not looked at
			bool maybe[LA_SEG];
			if(g_kbits) {
				uint32_t wd[LA_SEG];
#pragma unroll
				for(int i = 0; i < LA_SEG; ++i) wd[i] = g_kbits[((keys[i] * KMAHIP_KBITS_MUL) >> kb_sh) >> 5];
#pragma unroll
				for(int i = 0; i < LA_SEG; ++i) maybe[i] = (wd[i] >> (((keys[i] * KMAHIP_KBITS_MUL) >> kb_sh) & 31u)) & 1u;
			} else {
#pragma unroll
				for(int i = 0; i < LA_SEG; ++i) maybe[i] = true;
			}
			Bucket bk[LA_SEG];
#pragma unroll
			for(int i = 0; i < LA_SEG; ++i) {
				maybe[i] = maybe[i] && j0 + i < npos;
				const uint32_t b = maybe[i] ? (keys[i] * 0x9E3779B1u) >> g_sh : 0u;
				const KMAHIP_GLOBAL u32x4 *pb = g_slots + (size_t) b * (KMAHIP_BUCKET_SLOTS / 2);
				bk[i].a = pb[0]; bk[i].c = pb[1];
			}
#pragma unroll
			for(int i = 0; i < LA_SEG; ++i) res[i] = maybe[i] ? db_bucket_says(bk[i], keys[i]) : NOLIST;
#pragma unroll 1
			for(int i = 0; i < LA_SEG; ++i) if(res[i] == PROBE_MORE) {          // (the bucket is full of other keys: rare)
				uint32_t g2 = NOLIST;
#pragma unroll
				for(int y = 0; y < LA_SEG; ++y) if(y == i) g2 = db_probe_from(A.db, keys[y]);
#pragma unroll
				for(int y = 0; y < LA_SEG; ++y) if(y == i) res[y] = g2;
			}
			uint32_t lid[LA_SEG];
#pragma unroll
			for(int i = 0; i < LA_SEG; ++i) lid[i] = g_vs_id[res[i] != NOLIST ? res[i] : 0u];
#pragma unroll
			for(int i = 0; i < LA_SEG; ++i) v[i] = res[i] != NOLIST ? lid[i] : NOLIST;
		}
		// the last hit at or before each lane's segment (inclusive scan of "last non-empty"), then the one before the segment
		int lh = -1;
		uint32_t lv = NOLIST;
#pragma unroll
		for(int i = 0; i < LA_SEG; ++i) if(v[i] != NOLIST) { lh = j0 + i; lv = v[i]; }
		int h = lh;
		uint32_t hv = lv;
#pragma unroll
		for(int d = 1; d < 64; d <<= 1) {
			const int oh = __shfl_up(h, d);
			const uint32_t ov = __shfl_up(hv, d);
			if(lane >= d && h < 0) { h = oh; hv = ov; }
		}
		int ph = __shfl_up(h, 1);
		uint32_t pv = __shfl_up(hv, 1);
		if(lane == 0 || ph < 0) { ph = carry_h; pv = carry_v; }
		// open (1) / continue at once (2) / continue behind exactly k missed starts (3), per hit
		int opens = 0, th = ph;
		uint32_t tv = pv, code = 0;
#pragma unroll
		for(int i = 0; i < LA_SEG; ++i) {
			if(v[i] == NOLIST) continue;
			const int j = j0 + i;
			uint32_t c = 1;
			if(th >= 0 && v[i] == tv) { const int gaps = j - th - 1; if(gaps == 0) c = 2; else if(gaps == k) c = 3; }
			if(c == 1) ++opens;
			code |= c << (2 * i);
			th = j; tv = v[i];
		}
		int inc = opens;
#pragma unroll
		for(int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if(lane >= d) inc += o; }
		// slot of the anchor in hand where the segment starts: the anchors opened in this pass before it (0: the one carried in)
		int slot = inc - opens, accw = 0, accl = -1;
#pragma unroll
		for(int i = 0; i < LA_SEG; ++i) {
			const uint32_t c = (code >> (2 * i)) & 3u;
			if(!c) continue;
			const int j = j0 + i;
			if(c == 1) {
				if(accl >= 0) { atomicAdd(&s_w[slot], accw); atomicMax(&s_last[slot], (uint32_t) accl); }
				++slot;
				s_start[slot] = (uint32_t) j; s_val[slot] = v[i];
				accw = kM;
			} else accw += c == 2 ? M1 : kMM;
			accl = j;
		}
		if(accl >= 0) { atomicAdd(&s_w[slot], accw); atomicMax(&s_last[slot], (uint32_t) accl); }
		__syncthreads();
		const int n_new = __shfl(inc, 63);
		if(n_new > 0) {
			// every slot but the last is closed by the anchor behind it: it ends behind its last hit's k-mer + 1 (j - gaps + k at the hit
			// that opens the next one, savekmers.c:5290-5300)
			const int g0 = carry_cnt - 1;          // number of slot 0's anchor in the strand
			for(int s = lane; s < n_new; s += 64) {
				if(s == 0 && !carry_cnt) continue;
				CAnk a;
				a.score = 0; a.weight = s_w[s]; a.score_len = 0; a.len_len = 0;
				a.start = s_start[s]; a.end = s_last[s] + 1u + (uint32_t) k; a.values = s_val[s]; a.descend = g0 + s + 1;
				ank_store(&V[g0 + s], a);
			}
			const int cw = s_w[n_new];
			const uint32_t cl = s_last[n_new], cs = s_start[n_new], cv = s_val[n_new];
			__syncthreads();
			for(int s = lane; s <= n_new; s += 64) { s_w[s] = 0; s_last[s] = 0; }
			__syncthreads();
			if(lane == 0) { s_w[0] = cw; s_last[0] = cl; s_start[0] = cs; s_val[0] = cv; }
			__syncthreads();
			carry_cnt += n_new;
		}
		const int eh = __shfl(h, 63);
		const uint32_t ev = __shfl(hv, 63);
		if(eh >= 0) { carry_h = eh; carry_v = ev; }
	}
	if(lane == 0) {
		if(!carry_cnt) ank_store(&V[0], none);
		else {
			// the strand's last anchor ends at its last hit: seqlen - gaps with the k starts the read's end misses counted in (savekmers.c:5316-5330)
			CAnk a;
			a.score = 0; a.weight = s_w[0]; a.score_len = 0; a.len_len = 0;
			a.start = s_start[0]; a.end = s_last[0]; a.values = s_val[0]; a.descend = -1;
			ank_store(&V[carry_cnt - 1], a);
		}
		G.hits[2 * x + strand] = carry_cnt;
	}
}
// chaining and extraction of the reads whose anchors chain_long_anchor_kernel has built: a lane per read, per-template arrays DB_size wide
// in the lane's scratch as in chain_kernel
__global__ __launch_bounds__(64, CHAIN_MIN_WAVES) void chain_long_tail_kernel(const ChainArgs A, const LongArgs G) {
	const int64_t lane = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(lane >= A.lanes) return;
	uint8_t *base = A.scratch + lane * A.lane_bytes;
	const int64_t D = A.db.DB_size;
	CLaneT<DenseMap> L;
	L.db = &A.db;
	L.dbl.values_u16 = A.db.values_u16; L.dbl.mlen = A.db.mlen; L.dbl.values16 = A.db.values16; L.dbl.values32 = A.db.values32;
	L.VF = nullptr; L.VR = nullptr;
	L.tm.Score = (int *) base; base += (size_t) (D + 1) * 4;
	L.tm.extend = (int *) base; base += (size_t) (D + 1) * 4;
	L.bestT = (int *) base; base += (size_t) A.b_cap * 4;
	L.bestT_r = (int *) base; base += (size_t) A.b_cap * 4;
	L.tree = (CSeg *) base; base += (size_t) A.s_cap * sizeof(CSeg);
	L.ovf = (int *) (L.tree + (A.s_cap - 1));
	*(GInt *) L.ovf = 0;
	L.tm.include = (int8_t *) base;
	L.tm.tlen = A.db.tlen;
	L.k = (int) A.db.kmersize; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1; L.Wl = A.Wl;
	L.a_cap = A.a_cap; L.b_cap = A.b_cap; L.s_cap = A.s_cap;
	L.status = 0; L.tree_n = 0;
	for(int64_t i = lane; i < G.count; i += A.lanes) {
		const int64_t x = G.first + i, r = G.list[x];
		const int hitF = G.hits[2 * x], hitR = G.hits[2 * x + 1];
		const int seqlen = A.len[r];
		if(seqlen < L.k || (!hitF && !hitR) || A.stop_after == 1) continue;
		L.VF = G.pool + (G.v_off[x] - G.v_off[G.first]);
		L.VR = L.VF + (seqlen - L.k + 2);
		chain_read_tail(L, A, r, seqlen, (unsigned) hitF, (unsigned) hitR);
		if(L.status) {
			atomicMax(&A.counters[1], 40ull);
			for(int64_t t = 0; t <= D; ++t) { L.tm.Score[t] = 0; L.tm.extend[t] = 0; L.tm.include[t] = 0; }
			L.status = 0;
		}
	}
}

// The fast path, second kernel: chaining and extraction for the reads whose anchors chain_anchor_kernel (scan.hip) has built -- a
// lane per read like chain_kernel, but the anchors come ready from a compact pool and the per-template state lives in a hashed
// table in LDS, so a read costs a few dozen scattered accesses instead of 1 600. Per lane in HBM: two dummy anchors (a strand
// without hits), the two template lists and the tree of covered stretches. A read that overflows the LDS table (which can only
// happen while its anchors are chained, before anything has been written for it) is put on the list of the lane-per-read kernel.
struct FastArgs {
	CAnk *pool;
	const int64_t *a_off;
	const int32_t *a_n;
	uint8_t *slow;
	const uint32_t *order;    // the chunk's reads by falling number of anchors (or NULL: as they come): the 64 lanes of a wavefront pay for
	                          // the longest of their reads, so reads of a kind go together
};
// key = anchors on both strands (reads the kernel skips: 0, they end up together at the far end)
__global__ __launch_bounds__(256) void chain_order_keys_kernel(int64_t m, const int32_t *a_n, const uint8_t *slow, uint32_t *keys, uint32_t *vals) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= m) return;
	const int a = slow[r] ? 0 : a_n[2 * r] + a_n[2 * r + 1];
	keys[r] = (uint32_t) min(a, 255); vals[r] = (uint32_t) r;
}
#ifndef CHAIN_FAST_WAVES
#define CHAIN_FAST_WAVES 8
#endif
__global__ __launch_bounds__(64, CHAIN_FAST_WAVES) void chain_fast_kernel(const ChainArgs A, const FastArgs F) {
	__shared__ uint32_t t_id[LDS_TS * 64];
	__shared__ int t_sc[LDS_TS * 64], t_tl[LDS_TS * 64];
	__shared__ uint16_t t_ex[LDS_TS * 64];
	__shared__ int8_t t_inc[LDS_TS * 64];
	const int64_t lane = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint8_t *base = A.scratch + lane * A.lane_bytes;
	CLaneT<LdsMap> L;
	L.db = &A.db;
	L.dbl.values_u16 = A.db.values_u16; L.dbl.mlen = A.db.mlen; L.dbl.values16 = A.db.values16; L.dbl.values32 = A.db.values32;
	CAnk *dummy = (CAnk *) base; base += 2 * sizeof(CAnk);
	L.bestT = (int *) base; base += (size_t) A.b_cap * 4;
	L.bestT_r = (int *) base; base += (size_t) A.b_cap * 4;
	L.tree = (CSeg *) base;
	L.ovf = (int *) (L.tree + (A.s_cap - 1));          // (the last node is never used: seg_grow stops two short of s_cap)
	*(GInt *) L.ovf = 0;
	L.tm.id = (KMAHIP_LDS uint32_t *) t_id; L.tm.sc = (KMAHIP_LDS int *) t_sc; L.tm.tl = (KMAHIP_LDS int *) t_tl; L.tm.ex = (KMAHIP_LDS uint16_t *) t_ex;
	L.tm.inc = (KMAHIP_LDS int8_t *) t_inc; L.tm.tlen = A.db.tlen; L.tm.lane = (int) threadIdx.x;
	L.tm.reset();
	L.k = (int) A.db.kmersize; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1; L.Wl = A.Wl;
	L.a_cap = A.a_cap; L.b_cap = A.b_cap; L.s_cap = A.s_cap;
	L.status = 0; L.tree_n = 0;
	for(int64_t i = lane; i < A.n_reads; i += A.lanes) {
		const int64_t r = F.order ? (int64_t) F.order[i] : i;
		if(F.slow[r]) continue;
		const int nF = F.a_n[2 * r], nR = F.a_n[2 * r + 1];
		if(!nF && !nR) continue;
		for(int x = 0; x < 2; ++x) { dummy[x].start = 0; dummy[x].end = 0; dummy[x].values = NOLIST; dummy[x].descend = -1; dummy[x].score = 0; }
		L.VF = nF ? F.pool + F.a_off[2 * r] : &dummy[0];
		L.VR = nR ? F.pool + F.a_off[2 * r + 1] : &dummy[1];
		chain_read_tail(L, A, r, A.len[r], (unsigned) nF, (unsigned) nR);
		if(L.status) { F.slow[r] = 2; L.status = 0; }
		L.tm.reset();
	}
}

// the reads the fast route left: those without N's from the front of `list` (count[0]: the long-read route takes them when `split`), the
// others from its end (count[1]: chain_kernel)
__global__ __launch_bounds__(256) void slow_list_kernel(int64_t n, const uint8_t *slow, const int64_t *N_off, int split, int64_t *list, unsigned long long *count) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= n || !slow[r]) return;
	if(split && N_off[r + 1] == N_off[r]) list[atomicAdd(&count[0], 1ull)] = r;
	else list[n - 1 - (int64_t) atomicAdd(&count[1], 1ull)] = r;
}
__global__ __launch_bounds__(256) void chain_long_keys_kernel(int64_t n_list, const int64_t *list, const int32_t *len, uint32_t *keys) {
	const int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(x < n_list) keys[x] = (uint32_t) len[list[x]];
}

}  // namespace

int kmahip_worker_streams(hipStream_t *out, int want);      // longtrace.hip: streams that were measured to run side by side
void *kmahip_devcache_take(size_t bytes, size_t *got);      // pipeline.hip: the large device blocks kept between runs
void kmahip_devcache_give(void *p, size_t bytes);
void kmahip_devcache_flush();

// ---- the launch, everything in HBM: `d` holds DEVICE pointers; rec (8 ints per record: read lo, read hi, ordinal within the read,
// rc_flag, emit_rc, q_start, q_end, number of templates), rec_T (first template of the record in T) and T are device buffers of
// rec_cap / T_cap entries, filled in no particular order. n_recs / n_T: what the batch needs (KMAHIP_EOVERFLOW when that is more).
// Two routes (KMAHIP_CHAIN=slow forces the second for every read; the tests compare them):
//   fast   prefilter + chain_anchor_kernel (scan.hip: the anchors of every live strand, 16 lanes per strand) -> chain_fast_kernel
//          (a lane per read on ready anchors, per-template state in LDS), chunks of 2 M reads through one anchor pool;
//   slow   chain_kernel, a lane per read on DB_size-wide scratch, for the reads the fast route does not take (N's, more than 288
//          k-mer starts, more than 64 anchors on a strand, more candidate templates than the LDS table holds). -------------------
int kmahip_chain_device(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *d, const kmahip_params *p, const kmahip_chain_params *cp, int32_t *rec,
                        int64_t *rec_T, int32_t *T, int64_t rec_cap, int64_t T_cap, int64_t *n_recs, int64_t *n_T) {
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
		(void) hipDeviceSynchronize();
		const auto now = std::chrono::steady_clock::now();
		fprintf(stderr, "[kmahip] scan_chain: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	// (blocks of 64 MB and more come from, and go back to, the blocks the runs keep between calls -- pipeline.hip: a hipMalloc of
	// gigabytes takes a second now and then, and this function takes its pools anew for every batch)
	constexpr size_t KEEP = 64u << 20;
	std::vector<std::pair<void *, size_t>> owned;
	auto release = [](void *q, size_t b) { if(b >= KEEP) kmahip_devcache_give(q, b); else (void) hipFree(q); };
	struct Free {
		std::vector<std::pair<void *, size_t>> &v;
		~Free() { bool any = false; for(auto &q : v) any = any || q.second >= KEEP; if(any) (void) hipDeviceSynchronize(); for(auto &q : v) { if(q.second >= KEEP) kmahip_devcache_give(q.first, q.second); else (void) hipFree(q.first); } }
	} guard{owned};
	auto dev = [&](size_t bytes, void **out) -> int {
		const size_t want = bytes ? bytes : 16;
		size_t got = want;
		*out = want >= KEEP ? kmahip_devcache_take(want, &got) : nullptr;
		if(!*out) {
			got = want;
			if(hipMalloc(out, want) != hipSuccess) {
				kmahip_devcache_flush();
				if(hipMalloc(out, want) != hipSuccess) { *out = nullptr; kmahip_set_error("hipMalloc of %zu bytes failed", bytes); return KMAHIP_ENOMEM; }
			}
		}
		owned.push_back({*out, got});
		return KMAHIP_OK;
	};
	auto drop = [&](void *q) {
		for(size_t i = 0; i < owned.size(); ++i) if(owned[i].first == q) {
			if(owned[i].second >= KEEP) (void) hipDeviceSynchronize();
			release(q, owned[i].second);
			owned.erase(owned.begin() + (ptrdiff_t) i);
			return;
		}
	};
	int rc;
	ChainArgs A;
	A.db = db->dev; A.n_reads = n;
	A.seq = d->seq; A.seq_off = d->seq_off; A.len = d->len; A.N = d->N; A.N_off = d->N_off;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1; A.Wl = p->rw.Wl;
	A.stop_after = getenv("KMAHIP_CHAIN_STOP") ? atoi(getenv("KMAHIP_CHAIN_STOP")) : 0;
	A.exhaustive = p->exhaustive; A.minlen = cp ? cp->minlen : 16; A.coverT = cp ? cp->coverT : 0.1; A.mrs = cp ? cp->mrs : 0.5;
	A.read_base = 0;
	unsigned long long *counters = nullptr;
	if((rc = dev(KMAHIP_N_COUNTERS * 8, (void **) &counters))) return rc;
	HIP_TRY(hipMemsetAsync(counters, 0, KMAHIP_N_COUNTERS * 8, 0));
	A.counters = counters;
	A.rec = rec; A.rec_T = rec_T; A.T = T; A.rec_cap = rec_cap; A.T_cap = T_cap;
	// (tree nodes: two per accepted chain, and a chain is minlen bases at least; 128 = 63 chains cover every read up to 1 kb)
	const int s_cap_of_len = std::max(128, 2 * (max_len / std::max(A.minlen, 8)) + 8);

	const char *route = getenv("KMAHIP_CHAIN");
	const bool all_slow = route && !strcmp(route, "slow");
	uint8_t *slow = nullptr;
	int64_t *slow_list = nullptr, *long_list = nullptr;
	int64_t n_slow = all_slow ? n : 0, n_long = 0;
	if(!all_slow) {
		// (KMAHIP_CHAIN_CHUNK: reads per chunk, for the tests: many chunks out of a few thousand reads)
		const int64_t CHUNK = getenv("KMAHIP_CHAIN_CHUNK") ? std::max<int64_t>(64, atoll(getenv("KMAHIP_CHAIN_CHUNK"))) : 2000000;
		const int64_t m_max = std::min(n, CHUNK), n_chunks = (n + CHUNK - 1) / CHUNK;
		// The anchors of chunk i + 1 are made (on a stream of their own, into a second set of buffers) while chunk i is chained: the two
		// kernels wait for different things -- index lookups there, a lane's serial walk here. KMAHIP_CHAIN_OVERLAP=0, one chunk or
		// KMAHIP_DEBUG_TIMING (stamps per kernel): one set of buffers, one after the other.
		const int nbuf = (n_chunks > 1 && !dbg && !(getenv("KMAHIP_CHAIN_OVERLAP") && !atoi(getenv("KMAHIP_CHAIN_OVERLAP")))) ? 2 : 1;
		struct Buf { CAnk *pool = nullptr; int64_t pool_cap = 0; int64_t *a_off = nullptr; int32_t *a_n = nullptr; unsigned long long *cnt = nullptr; } buf[2];
		// (two of the process's worker streams, which were measured to run side by side: the chaining on the first, the anchors on the
		// second. On the null stream and a stream made here the two shared a hardware queue or not by what the process had done before.)
		struct Side {
			hipStream_t s = nullptr, m = nullptr; hipEvent_t e[2] = {nullptr, nullptr}; bool rec[2] = {false, false};
			~Side() { if(s) (void) hipStreamSynchronize(s); if(m) (void) hipStreamSynchronize(m); for(int x = 0; x < 2; ++x) if(e[x]) (void) hipEventDestroy(e[x]); }
		} side;
		{ hipStream_t W[2]; if((rc = kmahip_worker_streams(W, 2))) return rc; side.m = W[0]; side.s = W[1]; }
		for(int x = 0; x < 2; ++x) HIP_TRY(hipEventCreateWithFlags(&side.e[x], hipEventDisableTiming));
		if((rc = dev((size_t) n, (void **) &slow))) return rc;
		for(int x = 0; x < nbuf; ++x) {
			buf[x].pool_cap = m_max * 40 + 4096;
			if((rc = dev((size_t) m_max * 16, (void **) &buf[x].a_off)) || (rc = dev((size_t) m_max * 8, (void **) &buf[x].a_n)) || (rc = dev(16, (void **) &buf[x].cnt)) ||
			   (rc = dev((size_t) buf[x].pool_cap * sizeof(CAnk), (void **) &buf[x].pool))) return rc;
		}
		// per-lane scratch of the fast kernel: two dummy anchors, the two template lists, the tree
		ChainArgs Af = A;
		Af.a_cap = 0; Af.b_cap = 2 * LDS_TS + 8; Af.s_cap = 128;
		Af.lane_bytes = ((int64_t) 2 * (int64_t) sizeof(CAnk) + (int64_t) 2 * Af.b_cap * 4 + (int64_t) Af.s_cap * (int64_t) sizeof(CSeg) + 63) & ~63ll;
		Af.lanes = std::min<int64_t>(256 * 16 * 64, ((m_max + 63) / 64) * 64);
		void *fscratch = nullptr;
		if((rc = dev((size_t) (Af.lanes * Af.lane_bytes), &fscratch))) return rc;
		Af.scratch = (uint8_t *) fscratch;
		// (KMAHIP_CHAIN_ORDER=0: the reads as they come)
		const bool order_on = !(getenv("KMAHIP_CHAIN_ORDER") && !atoi(getenv("KMAHIP_CHAIN_ORDER")));
		uint32_t *o_keys = nullptr, *o_keys2 = nullptr, *o_vals = nullptr, *o_vals2 = nullptr;
		void *o_tmp = nullptr;
		size_t o_tmp_bytes = 0;
		if(order_on) {
			if(rocprim::radix_sort_pairs_desc((void *) nullptr, o_tmp_bytes, o_keys, o_keys2, o_vals, o_vals2, (size_t) m_max, 0u, 8u, (hipStream_t) 0) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs_desc (size query) failed"); return KMAHIP_EDEVICE; }
			if((rc = dev((size_t) m_max * 4, (void **) &o_keys)) || (rc = dev((size_t) m_max * 4, (void **) &o_keys2)) || (rc = dev((size_t) m_max * 4, (void **) &o_vals)) ||
			   (rc = dev((size_t) m_max * 4, (void **) &o_vals2)) || (rc = dev(std::max<size_t>(o_tmp_bytes, 16), &o_tmp))) return rc;
		}
		HIP_TRY(hipStreamSynchronize(0));          // (the counters are cleared before anything on the other stream starts)
		stamp("fast route: buffers");
		const auto t_route = std::chrono::steady_clock::now();
		auto chunk_view = [&](int64_t i) { kmahip_reads v = *d; const int64_t r0 = i * CHUNK; v.n_reads = std::min(CHUNK, n - r0); v.seq_off = d->seq_off + r0; v.len = d->len + r0; v.N_off = d->N_off + r0; return v; };
		// the anchors of chunk i into its set of buffers, once the chain kernel that read that set last is through
		auto launch_anchors = [&](int64_t i) -> int {
			const int x = (int) (i % nbuf);
			if(side.rec[x] && hipStreamWaitEvent(side.s, side.e[x], 0) != hipSuccess) { kmahip_set_error("hipStreamWaitEvent failed"); return KMAHIP_EDEVICE; }
			const kmahip_reads v = chunk_view(i);
			return kmahip_launch_chain_anchors(db, ws, &v, p, buf[x].pool, buf[x].pool_cap, buf[x].a_off, buf[x].a_n, slow + i * CHUNK, buf[x].cnt, side.s);
		};
		if((rc = launch_anchors(0))) return rc;
		for(int64_t i = 0; i < n_chunks; ++i) {
			const int x = (int) (i % nbuf);
			const int64_t r0 = i * CHUNK;
			const kmahip_reads v = chunk_view(i);
			const int64_t m = v.n_reads;
			for(int attempt = 0;; ++attempt) {
				unsigned long long used = 0;
				HIP_TRY(hipMemcpyAsync(&used, buf[x].cnt, 8, hipMemcpyDeviceToHost, side.s));
				HIP_TRY(hipStreamSynchronize(side.s));
				if((int64_t) used <= buf[x].pool_cap) break;
				if(attempt >= 2) { kmahip_set_error("anchor pool: %llu anchors for %lld reads", used, (long long) m); return KMAHIP_EOVERFLOW; }
				drop(buf[x].pool);          // (nothing reads it any more: launch_anchors waited for the chain kernel before this chunk's anchors)
				buf[x].pool_cap = (int64_t) used + 4096;
				if((rc = dev((size_t) buf[x].pool_cap * sizeof(CAnk), (void **) &buf[x].pool))) return rc;
				if((rc = launch_anchors(i))) return rc;
			}
			stamp("fast route: prefilter + chain_anchor_kernel");
			if(nbuf == 2 && i + 1 < n_chunks && (rc = launch_anchors(i + 1))) return rc;          // beside this chunk's chain kernel
			ChainArgs Ac = Af;
			Ac.n_reads = m; Ac.seq_off = v.seq_off; Ac.len = v.len; Ac.N_off = v.N_off; Ac.read_base = r0;
			FastArgs F = {buf[x].pool, buf[x].a_off, buf[x].a_n, slow + r0, nullptr};
			if(order_on) {
				hipLaunchKernelGGL(chain_order_keys_kernel, dim3((unsigned) ((m + 255) / 256)), dim3(256), 0, side.m, m, buf[x].a_n, slow + r0, o_keys, o_vals);
				if(rocprim::radix_sort_pairs_desc(o_tmp, o_tmp_bytes, o_keys, o_keys2, o_vals, o_vals2, (size_t) m, 0u, 8u, side.m) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs_desc failed"); return KMAHIP_EDEVICE; }
				F.order = o_vals2;
				stamp("fast route: reads ordered by their anchors");
			}
			hipLaunchKernelGGL(chain_fast_kernel, dim3((unsigned) (std::min<int64_t>(Af.lanes, ((m + 63) / 64) * 64) / 64)), dim3(64), 0, side.m, Ac, F);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipEventRecord(side.e[x], side.m));
			side.rec[x] = true;
			stamp("fast route: chain_fast_kernel");
			if(nbuf == 1 && i + 1 < n_chunks && (rc = launch_anchors(i + 1))) return rc;
		}
		HIP_TRY(hipStreamSynchronize(side.m));
		HIP_TRY(hipStreamSynchronize(side.s));
		if(getenv("KMAHIP_CHAIN_TIMING")) fprintf(stderr, "[kmahip] scan_chain: fast route, %lld reads in %lld chunks, %d set(s) of buffers: %.1f ms\n", (long long) n, (long long) n_chunks, nbuf,
		                                          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_route).count());
		unsigned long long *cnt = buf[0].cnt;
		for(int x = 0; x < nbuf; ++x) { drop(buf[x].pool); drop(buf[x].a_off); drop(buf[x].a_n); }
		drop(fscratch);
		// what is left for the lane-per-read kernel
		unsigned long long *cnt2 = cnt;
		if((rc = dev((size_t) n * 8, (void **) &slow_list))) return rc;
		HIP_TRY(hipMemsetAsync(cnt2, 0, 16, 0));
		// (KMAHIP_CHAIN_LONG=0: no long-read route, chain_kernel takes everything the fast route left)
		const int split = !(getenv("KMAHIP_CHAIN_LONG") && !atoi(getenv("KMAHIP_CHAIN_LONG")));
		hipLaunchKernelGGL(slow_list_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, 0, n, slow, d->N_off, split, slow_list, cnt2);
		unsigned long long ns[2] = {0, 0};
		HIP_TRY(hipMemcpy(ns, cnt2, 16, hipMemcpyDeviceToHost));
		n_long = (int64_t) ns[0];
		n_slow = (int64_t) ns[1];
		long_list = slow_list;
		slow_list += n - n_slow;
		if(dbg) fprintf(stderr, "[kmahip] scan_chain: of %lld reads %lld left to the long-read route, %lld to the lane-per-read kernel\n", (long long) n, (long long) n_long, (long long) n_slow);
	}
	if(n_long > 0) {
		// ---- the long-read route: chain_long_anchor_kernel (a wavefront per read and strand) + chain_long_tail_kernel (a lane per read) ----
		const int k = (int) db->info.kmersize;
		// the reads by falling length: the 64 lanes of a tail wavefront pay for the longest of their reads
		uint32_t *l_keys = nullptr, *l_keys2 = nullptr;
		int64_t *l_list2 = nullptr, *l_cnt = nullptr, *v_off = nullptr;
		int32_t *l_hits = nullptr;
		void *l_tmp = nullptr;
		size_t tmp_a = 0, tmp_b = 0;
		if(rocprim::radix_sort_pairs_desc((void *) nullptr, tmp_a, l_keys, l_keys2, long_list, l_list2, (size_t) n_long, 0u, 32u, (hipStream_t) 0) != hipSuccess ||
		   rocprim::exclusive_scan((void *) nullptr, tmp_b, l_cnt, v_off, (int64_t) 0, (size_t) n_long + 1, rocprim::plus<int64_t>(), (hipStream_t) 0) != hipSuccess) { kmahip_set_error("rocprim size query failed"); return KMAHIP_EDEVICE; }
		if((rc = dev((size_t) n_long * 4, (void **) &l_keys)) || (rc = dev((size_t) n_long * 4, (void **) &l_keys2)) || (rc = dev((size_t) n_long * 8, (void **) &l_list2)) ||
		   (rc = dev((size_t) (n_long + 1) * 8, (void **) &l_cnt)) || (rc = dev((size_t) (n_long + 1) * 8, (void **) &v_off)) || (rc = dev((size_t) n_long * 8, (void **) &l_hits)) ||
		   (rc = dev(std::max<size_t>(std::max(tmp_a, tmp_b), 16), &l_tmp))) return rc;
		const unsigned gl = (unsigned) ((n_long + 255) / 256);
		hipLaunchKernelGGL(chain_long_keys_kernel, dim3(gl), dim3(256), 0, 0, n_long, (const int64_t *) long_list, d->len, l_keys);
		if(rocprim::radix_sort_pairs_desc(l_tmp, tmp_a, l_keys, l_keys2, long_list, l_list2, (size_t) n_long, 0u, 32u, (hipStream_t) 0) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs_desc failed"); return KMAHIP_EDEVICE; }
		HIP_TRY(hipMemsetAsync(l_cnt + n_long, 0, 8, 0));
		hipLaunchKernelGGL(chain_long_sizes_kernel, dim3(gl), dim3(256), 0, 0, n_long, (const int64_t *) l_list2, d->len, k, l_cnt);
		if(rocprim::exclusive_scan(l_tmp, tmp_b, l_cnt, v_off, (int64_t) 0, (size_t) n_long + 1, rocprim::plus<int64_t>(), (hipStream_t) 0) != hipSuccess) { kmahip_set_error("rocprim::exclusive_scan failed"); return KMAHIP_EDEVICE; }
		int64_t total = 0;
		HIP_TRY(hipMemcpy(&total, v_off + n_long, 8, hipMemcpyDeviceToHost));
		// chunks of reads whose anchor regions fit the pool (KMAHIP_CHAIN_LONG_POOL_MB: 8 192; a read on its own always fits)
		const int64_t budget = std::max<int64_t>(1, getenv("KMAHIP_CHAIN_LONG_POOL_MB") ? atoll(getenv("KMAHIP_CHAIN_LONG_POOL_MB")) : 8192) * ((1ll << 20) / (int64_t) sizeof(CAnk));
		std::vector<int64_t> cut = {0, n_long};
		int64_t pool_n = total;
		if(total > budget) {
			std::vector<int64_t> h_off((size_t) n_long + 1);
			HIP_TRY(hipMemcpy(h_off.data(), v_off, (size_t) (n_long + 1) * 8, hipMemcpyDeviceToHost));
			cut.assign(1, 0);
			pool_n = 0;
			while(cut.back() < n_long) {
				const int64_t a = cut.back();
				int64_t b = (int64_t) (std::upper_bound(h_off.begin() + a, h_off.end(), h_off[(size_t) a] + budget) - h_off.begin()) - 1;
				b = std::min(std::max(b, a + 1), n_long);
				pool_n = std::max(pool_n, h_off[(size_t) b] - h_off[(size_t) a]);
				cut.push_back(b);
			}
		}
		CAnk *l_pool = nullptr;
		if((rc = dev((size_t) pool_n * sizeof(CAnk), (void **) &l_pool))) return rc;
		ChainArgs Al = A;
		Al.a_cap = 0; Al.b_cap = (int) std::min<int64_t>(2 * D + 4, 1 << 22); Al.s_cap = s_cap_of_len;
		Al.lane_bytes = ((D + 1) * 8 + (int64_t) 2 * Al.b_cap * 4 + (int64_t) Al.s_cap * (int64_t) sizeof(CSeg) + (D + 1) + 63) & ~63ll;
		int64_t lanes = 65536;
		while(lanes > 64 && lanes * Al.lane_bytes > (16ll << 30)) lanes >>= 1;
		int64_t most = 0;
		for(size_t c = 0; c + 1 < cut.size(); ++c) most = std::max(most, cut[c + 1] - cut[c]);
		lanes = std::min<int64_t>(lanes, ((most + 63) / 64) * 64);
		Al.lanes = lanes;
		void *l_scratch = nullptr;
		if((rc = dev((size_t) (lanes * Al.lane_bytes), &l_scratch))) return rc;
		HIP_TRY(hipMemsetAsync(l_scratch, 0, (size_t) (lanes * Al.lane_bytes), 0));
		Al.scratch = (uint8_t *) l_scratch;
		stamp("long-read route: lists, offsets, buffers");
		HIP_TRY(hipStreamSynchronize(0));
		const auto t_long = std::chrono::steady_clock::now();
		for(size_t c = 0; c + 1 < cut.size(); ++c) {
			LongArgs G = {l_list2, cut[c], cut[c + 1] - cut[c], v_off, l_pool, l_hits};
			hipLaunchKernelGGL(chain_long_anchor_kernel, dim3((unsigned) (2 * G.count)), dim3(64), 0, 0, Al, G);
			HIP_TRY(hipGetLastError());
			stamp("long-read route: chain_long_anchor_kernel");
			ChainArgs At = Al;
			At.lanes = std::min<int64_t>(lanes, ((G.count + 63) / 64) * 64);
			hipLaunchKernelGGL(chain_long_tail_kernel, dim3((unsigned) (At.lanes / 64)), dim3(64), 0, 0, At, G);
			HIP_TRY(hipGetLastError());
			stamp("long-read route: chain_long_tail_kernel");
		}
		HIP_TRY(hipStreamSynchronize(0));
		if(getenv("KMAHIP_CHAIN_TIMING")) fprintf(stderr, "[kmahip] scan_chain: long-read route, %lld reads in %zu chunk(s): %.2f ms (anchors + chaining; lists and buffers before it not counted)\n", (long long) n_long,
		                                          cut.size() - 1, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_long).count());
		drop(l_pool); drop(l_scratch);
	}
	if(n_slow > 0) {
		// every list may name all templates of the database (redundant databases: thousands share a k-mer), on either strand
		A.a_cap = max_len + 4; A.b_cap = (int) std::min<int64_t>(2 * D + 4, 1 << 22); A.s_cap = s_cap_of_len;
		A.lane_bytes = ((int64_t) 2 * A.a_cap * (int64_t) sizeof(CAnk) + (D + 1) * 8 + (int64_t) 2 * A.b_cap * 4 + (int64_t) A.s_cap * (int64_t) sizeof(CSeg) + (D + 1) + 63) & ~63ll;
		// 254 VGPRs: one wave per SIMD = 65 536 lanes resident. (Capped at 128 VGPRs for four waves per SIMD the kernel spills 1 000
		// registers and takes as long.)
		int64_t lanes = getenv("KMAHIP_CHAIN_LANES") ? atoll(getenv("KMAHIP_CHAIN_LANES")) : 65536;
		while(lanes > 64 && lanes * A.lane_bytes > (16ll << 30)) lanes >>= 1;
		lanes = std::min<int64_t>(lanes, ((n_slow + 63) / 64) * 64);
		A.lanes = lanes;
		void *scratch = nullptr;
		if((rc = dev((size_t) (lanes * A.lane_bytes), &scratch))) return rc;
		HIP_TRY(hipMemsetAsync(scratch, 0, (size_t) (lanes * A.lane_bytes), 0));
		A.scratch = (uint8_t *) scratch;
		stamp("slow route: scratch allocated and cleared");
		hipLaunchKernelGGL(chain_kernel, dim3((unsigned) (lanes / 64)), dim3(64), 0, 0, A, (const int64_t *) slow_list, n_slow);
		HIP_TRY(hipGetLastError());
		stamp("slow route: chain_kernel");
	}
	HIP_TRY(hipDeviceSynchronize());
	unsigned long long c[3] = {0, 0, 0};
	HIP_TRY(hipMemcpy(c, A.counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(&c[2], A.counters + CH_T, sizeof(unsigned long long), hipMemcpyDeviceToHost));
	*n_recs = (int64_t) c[0]; *n_T = (int64_t) c[2];
	if(c[1] == 40) { kmahip_set_error("default template finder: a per-read capacity ran out (more than %d chains in a read, or chains nested deeper than %d in the tree of covered stretches)", s_cap_of_len / 2, SEG_DEPTH); return KMAHIP_EOVERFLOW; }
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
	if((rc = kmahip_chain_device(db, ws, &d, p, cp, d_rec, d_rec_T, d_T, out->rec_cap, out->T_cap, &out->n_recs, &out->n_T))) return rc;
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
