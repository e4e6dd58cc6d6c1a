// assemble.hip -- stage 3c per template on gfx950: the pile-up of the traced reads (alnToMat, assembly.c:1317-1444, the
// default sparse matrix with insertion columns chained between template positions) on the device, and the consensus
// call + `.res` columns (callConsensus assembly.c:1499-1631, baseCaller :162-179, runkma.c:792-809) in host arithmetic
// (libm erf / tgamma exactly like the reference's p_chisqr).
//
// What makes the reference's pile-up sequential is only the insertion columns: a read that inserts bases the matrix has
// no column for creates columns whose gap count starts at the depth seen SO FAR (myBias, :1377-1397). Everything else is
// a commutative saturating increment. So: reads are sorted into the reference's order per template (ConClave prepends to
// a per-template list, conclave.c:164-165 -> reverse stream order inside every chunk of maxFrag records); one workgroup
// owns one template and goes through its reads 1024 at a time: maximal stretches of reads without an insertion run are
// piled up by all threads at once (atomics), a read with an insertion run is piled up alone, in order.
#include "kmahip_internal.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstdlib>

namespace {

struct InsNode { uint32_t c[6]; int32_t next; int32_t gaps; };   // next: following column of the chain (0 = none), gaps: template position after the chain

// a read's visit to a unit, ready to use: the unit's workgroups take their reads one after the other, and what a visit needs first --
// the read's figures, its checkpoint, where its runs and bases lie: four levels of dependent loads -- is gathered for all visits at
// once by pile_desc_kernel, a thread per visit; the workgroup then loads one record (the next visit's, while it works on this one)
struct VisitDesc {
	int64_t o;                   // first run of the read in `ops`
	int64_t qw;                  // first word of the read in `seq`
	int64_t Noff;                // first N position in `N`
	int32_t n, first;            // runs [first, n) are piled up (gap runs at either end trimmed)
	int32_t j_begin;             // run to start from, with the template columns / read bases in front of it:
	int32_t col_carry, q_carry;
	int32_t start;               // first template column of run `first`
	int32_t L, nN;
	int32_t flags;               // 1: reverse complement, 2: runs behind the unit need not be looked at
	int32_t pad;
};

static_assert(sizeof(VisitDesc) == 64, "a visit record is sixteen words");

struct PileArgs {
	DevDB db;
	int64_t n_reads;
	const uint64_t *seq;
	const int64_t *seq_off;
	const int32_t *len;
	const int32_t *N;
	const int64_t *N_off;
	const int32_t *flag, *tmpl;
	const int32_t *stats;        // 10 per read (kmahip_traces)
	const int64_t *ops_off;
	const int32_t *n_ops;
	const uint32_t *ops;
	int64_t max_frag;
	int order;                   // 0: ConClave's order (chunks of max_frag, each back to front), 1: stream order (`-Mt1`)
	int64_t *rank;               // per read: number of filed fragments before it in the stream
	// sorted work list
	uint64_t *keys;
	int32_t *vals;
	unsigned long long *counters;   // [0] kept reads, [1] status, [2] node pool top
	// output
	uint32_t *counts;            // 6 per position of `cat`
	int32_t *chain_head;         // per position of `cat`: first insertion column in front of it (0 = none), node ids are 1-based
	InsNode *nodes;
	int64_t node_cap;
	int32_t *seg_start;          // per template: 0 if it has kept reads, INT32_MAX if none (the consensus kernels ask); DB_size + 1
	int lds_node_limit;          // test hook (KMAHIP_PILE_LDS_NODES): fewer insertion columns per template in LDS than fit
	// work units: a template that fits LDS is one unit; a longer one is cut into segments of seg_cols columns, each piled up by
	// its own workgroup from the reads that cover it (a read visits every segment it touches)
	const int32_t *unit_base;    // DB_size + 1: first unit of template t
	const int32_t *unit_t;       // n_units: template of the unit
	const int32_t *unit_lo;      // n_units: first column of the unit
	int64_t n_units;
	int seg_cols;
	int64_t *vis_cnt, *vis_off;  // per read: number of units it visits, exclusive scan of that
	// per entry (in the order pile_keys_kernel lists them: vis_off[read] + unit - first unit visited): where in the read's runs a
	// unit's workgroup may start -- the last run that begins at or before the column two in front of the unit -- with the template
	// columns and read bases before that run (pile_ckpt_kernel). Without it every one of the ~10 units a 10 kb read touches scans
	// all of its ~3 000 runs.
	int32_t *ck_j, *ck_col, *ck_q;
	struct VisitDesc *desc;      // per sorted entry of a unit of a cut template: what its workgroup needs to start on the read (pile_desc_kernel)
	int64_t *unit_start;         // n_units + 1: first entry of the sorted list per unit (n_entries if none)
	int64_t *unit_end;           // n_units: one past its last entry (read only where unit_start < n_entries)
	int lds_words;               // LDS words a workgroup has (PILE_LDS_WORDS, or half of it so that two workgroups share a CU)
};

// oriented read base (0-3, 4 = N): the read as ConClave filed it
struct Q { const uint64_t *w; const int32_t *N; int L, nN, rc; int cw; uint64_t cv; };     // cw / cv: the word read last
__device__ __forceinline__ int q_base(Q &q, int i) {
	const int p = q.rc ? q.L - 1 - i : i;
	if(q.nN) {
		int lo = 0, hi = q.nN;
		while(lo < hi) { const int mid = (lo + hi) >> 1; if(q.N[mid] < p) lo = mid + 1; else hi = mid; }
		if(lo < q.nN && q.N[lo] == p) return 4;
	}
	// one load per 32 bases, not per base: the increments in between are atomics the compiler will not move a load across
	if((p >> 5) != q.cw) { q.cw = p >> 5; q.cv = q.w[p >> 5]; }
	const int b = (int) ((q.cv >> (62 - ((p & 31) << 1))) & 3ull);
	return q.rc ? 3 - b : b;
}

__global__ __launch_bounds__(256) void pile_filed_kernel(const int32_t *tmpl, int64_t n, int64_t *filed) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r < n) filed[r] = tmpl[r] != 0;
}

// Everything a template's workgroup shares lives in HBM and is read back by other threads of the SAME workgroup after a
// barrier. The increments are atomics (performed in L2); the few plain loads that must see them -- chain links, column
// depths -- go to L2 as well (agent-scope relaxed atomic loads) instead of invalidating the caches: a device-scope
// __threadfence() per phase writes back and invalidates the XCD's whole L2 and cost 0.5 ms a piece.
template <class T> __device__ __forceinline__ T ld_l2(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// chain links (plain stores by a thread of the same workgroup, read after a barrier): the CU's own L1 is coherent for them
template <class T> __device__ __forceinline__ T ld_wg(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// A template whose columns fit keeps EVERYTHING its workgroup shares in LDS while the reads are piled up -- counts (6 words a
// column), chain heads (1), and the insertion columns themselves (8 words each, numbered per template) -- and writes it out
// once at the end (the insertion columns get a contiguous range of the HBM pool then). The pile-up is bound by the increments
// (150 per read) and by walking the chains: L2 performs a few G atomics/s for the whole chip and a chain link read from it
// takes a microsecond, where every CU's LDS does its own in tens of cycles. Templates too long for that (and a launch whose
// insertion columns overflowed the LDS table, which is repeated) work on HBM as described above.
extern __shared__ uint32_t pile_lds[];      // [6 * t_len] counts, [t_len] chain heads, [8 * lds_nodes] insertion columns
constexpr int PILE_LDS_WORDS = (160 * 1024 - 1024) / 4;
constexpr int PILE_LDS_MIN_NODES = 64;
constexpr int PILE_THREADS = 1024;     // one workgroup per unit: reads are taken a workgroup's threads at a time (short reads) or one by one
                                       // (segments). Launched with PILE_THREADS threads and all of the LDS, or with half of both: two per CU

template <bool LDS>
struct Walk {
	const PileArgs &A;
	int64_t tbase;
	int t_len;
	unsigned *s_nodes;          // LDS mode: insertion columns in use
	__device__ __forceinline__ int node_base() const { return 7 * t_len; }
	__device__ __forceinline__ int node_cap() const { return min((A.lds_words - 7 * t_len) / 8, A.lds_node_limit); }
	// the chain head in front of template position p (1-based column ids, 0 = none)
	__device__ __forceinline__ int head(int p) const { return LDS ? (int) pile_lds[6 * t_len + p] : ld_wg(&A.chain_head[tbase + p]); }
	__device__ __forceinline__ void set_head(int p, int id) const { if(LDS) pile_lds[6 * t_len + p] = (uint32_t) id; else A.chain_head[tbase + p] = id; }
	// insertion column h
	__device__ __forceinline__ int next(int h) const { return LDS ? (int) pile_lds[node_base() + 8 * (h - 1) + 6] : ld_wg(&A.nodes[h - 1].next); }
	__device__ __forceinline__ void set_next(int h, int id) const { if(LDS) pile_lds[node_base() + 8 * (h - 1) + 6] = (uint32_t) id; else A.nodes[h - 1].next = id; }
	__device__ __forceinline__ void add_node(int h, int b) const {
		if(LDS) atomicAdd(&pile_lds[node_base() + 8 * (h - 1) + b], 1u); else atomicAdd(&A.nodes[h - 1].c[b], 1u);
	}
	__device__ __forceinline__ void add_col(int p, int b) const {      // template column p
		if(LDS) atomicAdd(&pile_lds[6 * p + b], 1u); else atomicAdd(&A.counts[6 * (tbase + p) + b], 1u);
	}
	// depth as the reference's 16-bit counters hold it; h = 0: template column p
	__device__ int depth16(int h, int p) const {
		int s = 0;
		if(LDS) { const uint32_t *c = h ? &pile_lds[node_base() + 8 * (h - 1)] : &pile_lds[6 * p]; for(int j = 0; j < 6; ++j) s += (int) min(c[j], 65535u); }
		else { const uint32_t *c = h ? A.nodes[h - 1].c : A.counts + 6 * (tbase + p); for(int j = 0; j < 6; ++j) s += (int) min(ld_l2(&c[j]), 65535u); }
		return s;
	}
	// a new insertion column (called by one thread at a time); 0: none left (status is set)
	__device__ int new_node(int bias, int b, int gaps) const {
		if(LDS) {
			const int id = (int) ++*s_nodes;
			if(id > node_cap()) { atomicMax(&A.counters[1], 16ull); return 0; }
			uint32_t *c = &pile_lds[node_base() + 8 * (id - 1)];
			for(int x = 0; x < 6; ++x) c[x] = 0;
			c[5] = (uint32_t) bias; c[b] = 1; c[6] = 0; c[7] = (uint32_t) gaps;
			return id;
		}
		const long long id = (long long) atomicAdd(&A.counters[2], 1ull) + 1;
		if(id > A.node_cap) { atomicMax(&A.counters[1], 32ull); return 0; }
		InsNode &nn = A.nodes[id - 1];
		for(int x = 0; x < 6; ++x) nn.c[x] = 0;
		nn.c[5] = (uint32_t) bias; nn.c[b] = 1; nn.next = 0; nn.gaps = gaps;
		return (int) id;
	}
};

// alnToMat for one read, split into what commutes and what does not (assembly.c:1317-1444). The runs of a kept read, after the
// leading / trailing gap runs are trimmed (assembly.c:1340-1354), cover the template columns start .. start + span - 1 of the
// ring: every column gets the read's base (or a gap), every insertion chain BETWEEN two covered columns a gap per column
// ("this read lacks them"). Those are plain increments and any number of lanes may do them side by side: pile_cols gives
// lane k of G a contiguous share of one read's columns. An insertion run -- bases for the chain in front of its column,
// new columns when the chain is too short, whose gap count starts at the depth seen so far -- is order dependent: pile_ins
// does these, one lane, after everything earlier in the order is in. To keep the depths it reads what the reference sees
// (the column before the site counted, the column behind it not yet), pile_cols leaves the column behind an insertion run
// and the chain in front of it to pile_ins.
struct ReadRuns {
	Q q;
	int64_t o;
	int n, first, start, qp;
	int lead_d, trail_d;      // template columns of the gap runs trimmed in front / behind
};
__device__ __forceinline__ ReadRuns read_runs(const PileArgs &A, int64_t r) {
	ReadRuns R;
	const int32_t *st = A.stats + 10 * r;
	R.o = A.ops_off[r];
	R.n = A.n_ops[r];
	R.q.w = A.seq + A.seq_off[r]; R.q.L = A.len[r]; R.q.N = A.N + A.N_off[r]; R.q.nN = (int) (A.N_off[r + 1] - A.N_off[r]);
	R.q.rc = (((A.flag[r] & 1) != 0) != (A.tmpl[r] < 0)) ? 1 : 0;
	R.q.cw = -1; R.q.cv = 0;
	R.start = st[1]; R.qp = st[4]; R.first = 0; R.lead_d = 0; R.trail_d = 0;
	// column 0 is never trimmed from the back
	while(R.n > 1 && (A.ops[R.o + R.n - 1] & 3u) >= 2u) { const uint32_t run = A.ops[R.o + R.n - 1]; if((run & 3u) == 3u) R.trail_d += (int) (run >> 2); --R.n; }
	while(R.first < R.n && (A.ops[R.o + R.first] & 3u) >= 2u) {
		const uint32_t run = A.ops[R.o + R.first];
		if((run & 3u) == 3u) { R.start += (int) (run >> 2); R.lead_d += (int) (run >> 2); } else R.qp += (int) (run >> 2);
		++R.first;
	}
	return R;
}

__device__ bool read_has_ins(const PileArgs &A, int64_t r) {
	const ReadRuns R = read_runs(A, r);
	bool ins = false;
	for(int j = R.first; j < R.n; ++j) if((A.ops[R.o + j] & 3u) == 2u) ins = true;
	return ins;
}

// the units a kept read visits: [u0, u1] and, for an alignment that wraps around a circular template, [w0, w1] as well. A unit
// [lo, hi) is visited when the read covers one of the columns lo - 1 .. hi - 1: the column in front of the unit is counted there
// too (its depth is what a new insertion column in front of column lo starts from, assembly.c:1377-1397).
struct Visits { int u0, u1, w0, w1; };
__device__ Visits pile_visits(const PileArgs &A, int64_t r) {
	Visits V = {0, -1, 0, -1};
	const int t = abs(A.tmpl[r]);
	const int ub = A.unit_base[t], nu = A.unit_base[t + 1] - ub;
	if(nu == 1) { V.u0 = V.u1 = ub; return V; }
	const ReadRuns R = read_runs(A, r);
	const int32_t *st = A.stats + 10 * r;
	const int t_len = A.db.tlen[t];
	const int span = st[3] - st[7] - R.lead_d - R.trail_d;       // template columns the piled-up part covers
	const int S = A.seg_cols;
	int start = R.start;
	if(start >= t_len) start -= t_len;
	const int end = start + span;                                 // exclusive; > t_len: wraps
	if(end <= t_len) {
		V.u0 = ub + start / S; V.u1 = ub + min(nu - 1, end / S);
		// (unit 0 also counts the last column of a circular template as the column in front of it)
		if(end == t_len && V.u0 > ub) { V.w0 = V.w1 = ub; }
	} else {
		V.u0 = ub + start / S; V.u1 = ub + nu - 1;
		V.w0 = ub; V.w1 = ub + min(nu - 1, (end - t_len) / S);
		if(V.w1 >= V.u0) { V.u0 = ub; V.w1 = -1; V.w0 = 0; }     // covers everything
	}
	return V;
}

__global__ __launch_bounds__(256) void pile_count_kernel(const PileArgs A) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= A.n_reads) return;
	int64_t c = 0;
	if(A.stats[10 * r + 3] != 0) {
		const Visits V = pile_visits(A, r);
		c = (V.u1 - V.u0 + 1) + (V.w1 >= V.w0 ? V.w1 - V.w0 + 1 : 0);
		A.seg_start[abs(A.tmpl[r])] = 0;
	}
	A.vis_cnt[r] = c;
}

// sort key: unit, then the order the reference assembles the reads of a template in
__global__ __launch_bounds__(256) void pile_keys_kernel(const PileArgs A) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= A.n_reads || A.stats[10 * r + 3] == 0) return;
	// rank among the fragments ConClave filed (conclave.c:166, 194) -> chunk of max_frag, reverse order inside the chunk
	const int64_t rk = A.rank[r];
	const uint64_t chunk = (uint64_t) (rk / A.max_frag), in = (uint64_t) (rk % A.max_frag);
	const uint64_t ord = A.order ? (uint64_t) rk : chunk * (uint64_t) A.max_frag + ((uint64_t) A.max_frag - 1 - in);
	const Visits V = pile_visits(A, r);
	int64_t slot = A.vis_off[r];
	for(int u = V.u0; u <= V.u1; ++u, ++slot) { A.keys[slot] = ((uint64_t) u << 28) | ord; A.vals[slot] = (int32_t) r; }
	for(int u = V.w0; u <= V.w1; ++u, ++slot) { A.keys[slot] = ((uint64_t) u << 28) | ord; A.vals[slot] = (int32_t) r; }
}

// checkpoints of the reads that lie on a template cut into units and do not wrap around its end: one wavefront per read, its runs
// 64 at a time (prefix sums of template columns / read bases by wave shuffles)
__global__ __launch_bounds__(64) void pile_ckpt_kernel(const PileArgs A) {
	const int lane = threadIdx.x;
	for(int64_t r = blockIdx.x; r < A.n_reads; r += gridDim.x) {
		if(A.stats[10 * r + 3] == 0) continue;
		const int t = abs(A.tmpl[r]);
		const int ub = A.unit_base[t], nu = A.unit_base[t + 1] - ub;
		if(nu <= 1) continue;
		const Visits V = pile_visits(A, r);
		const ReadRuns R = read_runs(A, r);
		const int t_len = A.db.tlen[t];
		int start = R.start;
		if(start >= t_len) start -= t_len;
		const int nvis = V.u1 - V.u0 + 1;
		const int64_t slot0 = A.vis_off[r];
		const int nslots = nvis + (V.w1 >= V.w0 ? V.w1 - V.w0 + 1 : 0);
		for(int x = lane; x < nslots; x += 64) { A.ck_j[slot0 + x] = R.first; A.ck_col[slot0 + x] = 0; A.ck_q[slot0 + x] = R.qp; }
		if(V.w1 >= V.w0 || V.u0 != ub + start / A.seg_cols) continue;          // (wraps, or covers everything: the units scan from the first run)
		const int S = A.seg_cols, u0l = V.u0 - ub;
		int col_carry = 0, q_carry = R.qp;
		for(int j0 = R.first; j0 < R.n; j0 += 64) {
			const int j = j0 + lane;
			const bool valid = j < R.n;
			const uint32_t run = valid ? A.ops[R.o + j] : 0u;
			const int cls = (int) (run & 3u), len = (int) (run >> 2);
			const int tl = valid && cls != 2 ? len : 0, ql = valid && cls != 3 ? len : 0;
			int ct = tl, cq = ql;
			for(int d = 1; d < 64; d <<= 1) { const int a = __shfl_up(ct, d), b = __shfl_up(cq, d); if(lane >= d) { ct += a; cq += b; } }
			const int col0 = col_carry + ct - tl, q0 = q_carry + cq - ql;
			// this run is the place to start for every unit whose column lo - 2 lies in [col0, col0 + tl - 1] (relative to the read's
			// first column); a run without template columns (an insertion) is never one
			if(tl > 0) {
				const int a0 = col0 + start + 2, a1 = col0 + tl - 1 + start + 2;          // lo of the units: a0 <= lo <= a1, lo a multiple of S
				for(int ui = (a0 + S - 1) / S - u0l; ui <= a1 / S - u0l && ui < nvis; ++ui)
					if(ui >= 0) { A.ck_j[slot0 + ui] = j; A.ck_col[slot0 + ui] = col0; A.ck_q[slot0 + ui] = q0; }
			}
			col_carry += __shfl(ct, 63); q_carry += __shfl(cq, 63);
		}
	}
}

__global__ __launch_bounds__(256) void pile_segments_kernel(const uint64_t *keys, int64_t n_ent, int64_t *unit_start, int64_t *unit_end) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n_ent) return;
	const int64_t u = (int64_t) (keys[i] >> 28);
	if(i == 0 || (int64_t) (keys[i - 1] >> 28) != u) unit_start[u] = i;
	if(i + 1 == n_ent || (int64_t) (keys[i + 1] >> 28) != u) unit_end[u] = i + 1;
}

// one thread per sorted entry: the visit's record (only for units of templates that are cut into units)
__global__ __launch_bounds__(256) void pile_desc_kernel(const PileArgs A, int64_t n_ent) {
	const int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(e >= n_ent) return;
	const int64_t r = A.vals[e];
	const int unit = (int) (A.keys[e] >> 28);
	const int t = abs(A.tmpl[r]);
	const int ub = A.unit_base[t];
	if(A.unit_base[t + 1] - ub <= 1) return;
	const ReadRuns R = read_runs(A, r);
	const int t_len = A.db.tlen[t];
	VisitDesc D;
	D.o = R.o; D.qw = A.seq_off[r]; D.Noff = A.N_off[r];
	D.n = R.n; D.first = R.first; D.L = R.q.L; D.nN = R.q.nN;
	int start = R.start;
	if(start >= t_len) start -= t_len;
	D.start = start;
	D.j_begin = R.first; D.col_carry = 0; D.q_carry = R.qp;
	D.flags = R.q.rc ? 1 : 0; D.pad = 0;
	// where this unit's stretch of the read begins (pile_ckpt_kernel); a read that wraps round the template's end has no checkpoints
	const int u0 = ub + start / A.seg_cols;
	const int32_t *st = A.stats + 10 * r;
	const int span = st[3] - st[7] - R.lead_d - R.trail_d;
	if(A.ck_j && start + span <= t_len && unit >= u0) {
		const int64_t slot = A.vis_off[r] + (unit - u0);
		D.j_begin = A.ck_j[slot]; D.col_carry = A.ck_col[slot]; D.q_carry = A.ck_q[slot];
		D.flags |= 2;
	}
	A.desc[e] = D;
}

__global__ __launch_bounds__(256) void pile_fill_kernel(int64_t *p, int64_t n, int64_t v) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i < n) p[i] = v;
}

// ---- a template too long for LDS: cut into units of seg_cols columns, one workgroup per unit, the reads that cover the unit in
// the reference's order, ONE READ AT A TIME with all 1024 threads on it (long noisy reads all carry insertions, so there is no
// stretch of insertion-free reads to pile up side by side; the parallelism is inside the read). Thread j takes run j of the read:
//   scan     block prefix sums of the runs' template / query lengths -> where every run starts
//   phase 1  runs of aligned pairs / gaps in the read: the columns they cover inside the unit (+ the column in front of it),
//            a gap for every insertion column chained in front of a covered column -- except the first column behind an
//            insertion run and its chain, which wait for ...
//   phase 2a ... the insertion runs, each on the chain in front of its column: bases into the existing columns, new columns
//            when the chain is too short, started from the depths the reference sees at that moment (the column in front of
//            the site counted, the one behind it not yet; two insertion runs one aligned base apart read each other's column,
//            hence the explicit `pending` base), then
//   phase 2b the column behind each insertion run.
struct SegWalk {
	const PileArgs &A;
	int lo, hi, t_len, ncol;        // columns [lo, hi) at index 1 .., index 0 = the column in front (lo - 1, or the last one of a ring)
	unsigned *s_nodes;
	__device__ __forceinline__ int ci(int p) const {
		if(p >= lo && p < hi) return p - lo + 1;
		return (p == (lo ? lo - 1 : t_len - 1)) ? 0 : -1;
	}
	__device__ __forceinline__ int node_base() const { return 7 * ncol; }
	// (the last 3 words per thread of the LDS hold the round's runs: first template column, first read base, the run itself)
	__device__ __forceinline__ int run_base() const { return A.lds_words - 3 * (int) blockDim.x; }
	__device__ __forceinline__ int node_cap() const { return min((run_base() - 7 * ncol) / 8, A.lds_node_limit); }
	__device__ __forceinline__ int head(int c) const { return (int) pile_lds[6 * ncol + c]; }
	__device__ __forceinline__ void set_head(int c, int id) const { pile_lds[6 * ncol + c] = (uint32_t) id; }
	__device__ __forceinline__ int next(int h) const { return (int) pile_lds[node_base() + 8 * (h - 1) + 6]; }
	__device__ __forceinline__ void set_next(int h, int id) const { pile_lds[node_base() + 8 * (h - 1) + 6] = (uint32_t) id; }
	__device__ __forceinline__ void add_node(int h, int b) const { atomicAdd(&pile_lds[node_base() + 8 * (h - 1) + b], 1u); }
	__device__ __forceinline__ void add_col(int c, int b) const { atomicAdd(&pile_lds[6 * c + b], 1u); }
	__device__ __forceinline__ int depth16_col(int c) const { int s = 0; for(int j = 0; j < 6; ++j) s += (int) min(pile_lds[6 * c + j], 65535u); return s; }
	__device__ __forceinline__ int depth16_node(int h) const { int s = 0; for(int j = 0; j < 6; ++j) s += (int) min(pile_lds[node_base() + 8 * (h - 1) + j], 65535u); return s; }
	__device__ int new_node(int bias, int b, int gaps) const {
		const int id = (int) atomicAdd(s_nodes, 1u) + 1;
		if(id > node_cap()) { atomicMax(&A.counters[1], 64ull); return 0; }
		uint32_t *c = &pile_lds[node_base() + 8 * (id - 1)];
		for(int x = 0; x < 6; ++x) c[x] = 0;
		c[5] = (uint32_t) bias; c[b] = 1; c[6] = 0; c[7] = (uint32_t) gaps;
		return id;
	}
};

// exclusive prefix sums over the workgroup of two lengths packed into one word (template columns high, query bases low); *total
// = their sums. s_w: one word per wavefront.
__device__ uint64_t block_scan2(uint64_t v, unsigned long long *s_w, uint64_t *total) {
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	uint64_t x = v;
	for(int d = 1; d < 64; d <<= 1) { const uint64_t y = __shfl_up(x, d); if(lane >= d) x += y; }
	__syncthreads();
	if(lane == 63) s_w[wave] = x;
	__syncthreads();
	uint64_t base = 0, tot = 0;
	for(int w = 0; w < (int) blockDim.x / 64; ++w) { const uint64_t t = s_w[w]; if(w < wave) base += t; tot += t; }
	*total = tot;
	return base + x - v;
}

__device__ __forceinline__ void pile_seg_read(const PileArgs &A, const SegWalk &W, const VisitDesc &V, uint32_t pre_run, uint32_t pre_prun, uint32_t pre_nrun, unsigned long long *s_w) {
	const int tid = threadIdx.x;
	ReadRuns R;
	R.o = V.o; R.n = V.n; R.first = V.first;
	R.q.w = A.seq + V.qw; R.q.L = V.L; R.q.N = A.N + V.Noff; R.q.nN = V.nN; R.q.rc = V.flags & 1; R.q.cw = -1; R.q.cv = 0;
	const int start = V.start;
	int64_t col_carry = V.col_carry, q_carry = V.q_carry;
	const int j_begin = V.j_begin;
	const bool ends_early = (V.flags & 2) != 0;
	for(int j0 = j_begin; j0 < R.n; j0 += (int) blockDim.x) {
		if(ends_early && start + col_carry > (int64_t) W.hi) break;          // (every further run lies behind the unit)
		const int j = j0 + tid;
		const bool valid = j < R.n;
		const bool pre = j0 == j_begin;                 // the first round's runs were asked for during the visit before
		const uint32_t run = valid ? (pre ? pre_run : A.ops[R.o + j]) : 0u;
		const int cls = (int) (run & 3u), len = (int) (run >> 2);
		const uint64_t mine = valid ? (((uint64_t) (cls != 2 ? len : 0) << 32) | (uint64_t) (cls != 3 ? len : 0)) : 0ull;
		uint64_t total;
		const uint64_t ex = block_scan2(mine, s_w, &total);
		const int64_t col0 = col_carry + (int64_t) (ex >> 32), q0 = q_carry + (int64_t) (ex & 0xFFFFFFFFull);
		col_carry += (int64_t) (total >> 32); q_carry += (int64_t) (total & 0xFFFFFFFFull);
		const uint32_t prun = (valid && j > R.first) ? (pre ? pre_prun : A.ops[R.o + j - 1]) : 0u;
		// ---- phase 1: the unit's columns the round's runs cover, a thread per COLUMN (a thread per run waits for the longest run: sixty
		// columns, each with its chain of insertion columns to walk). The runs' first columns / bases go to LDS, a column finds its run
		// by bisection. Column index ci: 0 = the column in front of the unit (the ring's last one for the first unit of a template
		// that is not one unit), 1 .. = lo ..
		{
			uint32_t *rc0 = pile_lds + W.run_base(), *rq0 = rc0 + blockDim.x, *rrun = rq0 + blockDim.x;
			const int64_t colA = col_carry - (int64_t) (total >> 32);
			rc0[tid] = valid ? (uint32_t) (col0 - colA) : 0xFFFFFFFFu;
			rq0[tid] = (uint32_t) q0;
			rrun[tid] = run;
			__syncthreads();
			const int nr = min((int) blockDim.x, R.n - j0);
			const uint32_t prev0 = j0 > R.first ? A.ops[R.o + j0 - 1] : 0u;       // the run in front of the round's first
			const int64_t span = col_carry - colA;
			for(int ci = tid; ci < W.ncol; ci += (int) blockDim.x) {
				int p;
				if(ci >= 1) p = W.lo + ci - 1;
				else if(W.lo) p = W.lo - 1;
				else if(W.hi < W.t_len) p = W.t_len - 1;
				else continue;
				int64_t rel = (int64_t) p - start;
				if(rel < 0) rel += W.t_len;
				rel -= colA;
				if(rel < 0 || rel >= span) continue;
				// last run that begins at or before the column (a run without template columns shares its column with the run behind it)
				int a = 0, bnd = nr;
				while(bnd - a > 1) { const int mid = (a + bnd) >> 1; if((int64_t) rc0[mid] <= rel) a = mid; else bnd = mid; }
				const uint32_t rj = rrun[a];
				const int rcls = (int) (rj & 3u);
				if(rcls == 2) continue;
				const int c = (int) (rel - (int64_t) rc0[a]);
				const uint32_t pj = a ? rrun[a - 1] : prev0;
				if(c == 0 && (j0 + a) > R.first && (pj & 3u) == 2u) continue;      // the first column behind an insertion run: phase 2b
				W.add_col(ci, rcls == 3 ? 5 : q_base(R.q, (int) ((int64_t) (int32_t) rq0[a] + c)));
				if(rel + colA > 0 && ci >= 1) for(int h = W.head(ci); h; h = W.next(h)) W.add_node(h, 5);
			}
		}
		__syncthreads();
		// ---- phase 2a: insertion runs whose column lies in the unit
		int site = -1, qafter = 0;
		if(valid && cls == 2) {
			const int p = (int) ((start + col0) % W.t_len);
			const int ci = W.ci(p);
			if(ci >= 0) { site = ci; qafter = (int) (q0 + len); }      // (index 0, the column in front of the unit: counted, its chain is not ours)
			if(ci >= 1) {
				int left = len, qpos = (int) q0, last = 0;
				int h = W.head(ci);
				while(h && left > 0) { W.add_node(h, q_base(R.q, qpos++)); last = h; h = W.next(h); --left; }
				if(left > 0) {
					int myBias;
					if(last) myBias = W.depth16_node(last);
					else {
						const int cprev = W.ci(p ? p - 1 : W.t_len - 1);
						myBias = W.depth16_col(cprev);
						// the column in front of the site is the single column between two insertion runs of this read: its own
						// base is added in phase 2b of this round -- count it now (16-bit counters: only while below the ceiling)
						if(j - 2 >= j0 && j - 2 >= R.first && (prun & 3u) != 2u && (prun >> 2) == 1u && (A.ops[R.o + j - 2] & 3u) == 2u) {
							const int b = (prun & 3u) == 3u ? 5 : q_base(R.q, (int) q0 - 1);
							if(pile_lds[6 * cprev + b] < 65535u) ++myBias;
						}
					}
					const int tmp = W.depth16_col(ci);
					myBias = (tmp < myBias) ? tmp : (myBias - 1);
					if(65535 < myBias) myBias = 65535;
					while(left > 0) {
						const int id = W.new_node(myBias, q_base(R.q, qpos++), p);
						if(!id) break;
						if(last) W.set_next(last, id); else W.set_head(ci, id);
						last = id;
						--left;
					}
				} else for(; h; h = W.next(h)) W.add_node(h, 5);
			}
		}
		__syncthreads();
		// ---- phase 2b: the column behind the insertion (first element of the next run; there is one: trailing gap runs are trimmed)
		if(site >= 0) {
			const int ncls = j + 1 < R.n ? (int) ((pre ? pre_nrun : A.ops[R.o + j + 1]) & 3u) : 0;
			W.add_col(site, ncls == 3 ? 5 : q_base(R.q, qafter));
		}
		__syncthreads();
	}
}

template <bool LDS>
__device__ void pile_cols(const PileArgs &A, const Walk<LDS> &W, int64_t r, int k, int G) {
	ReadRuns R = read_runs(A, r);
	int span = 0;
	for(int j = R.first; j < R.n; ++j) { const uint32_t run = A.ops[R.o + j]; if((run & 3u) != 2u) span += (int) (run >> 2); }
	const int chunk = (span + G - 1) / G, lo = k * chunk, hi = min(span, lo + chunk);
	if(lo >= hi) return;
	int i = 0, qpos = R.qp;
	bool after_ins = false;
	for(int j = R.first; j < R.n && i < hi; ++j) {
		const uint32_t run = A.ops[R.o + j];
		const int cls = (int) (run & 3u), len = (int) (run >> 2);
		if(cls == 2) { qpos += len; after_ins = true; continue; }
		for(int c = max(i, lo); c < min(i + len, hi); ++c) {
			if(after_ins && c == i) continue;                       // pile_ins
			int p = R.start + c;
			if(p >= W.t_len) p -= W.t_len;
			W.add_col(p, cls == 3 ? 5 : q_base(R.q, qpos + (c - i)));
			if(c > 0) for(int h = W.head(p); h; h = W.next(h)) W.add_node(h, 5);
		}
		if(cls != 3) qpos += len;
		i += len;
		after_ins = false;
	}
}

template <bool LDS>
__device__ void pile_ins(const PileArgs &A, const Walk<LDS> &W, int64_t r) {
	ReadRuns R = read_runs(A, r);
	int i = 0, qpos = R.qp;
	for(int j = R.first; j < R.n; ++j) {
		const uint32_t run = A.ops[R.o + j];
		const int cls = (int) (run & 3u);
		int left = (int) (run >> 2);
		if(cls != 2) { if(cls != 3) qpos += left; i += left; continue; }
		int gaps = R.start + i;                                     // the template column behind the insertion
		if(gaps >= W.t_len) gaps -= W.t_len;
		int last = 0;                                               // last column of the chain walked so far (0: none)
		int h = W.head(gaps);
		// insertion columns that already exist
		while(h && left > 0) {
			W.add_node(h, q_base(R.q, qpos++));
			last = h;
			h = W.next(h);
			--left;
		}
		if(left > 0) {
			// new columns in front of template position `gaps` (assembly.c:1368-1428)
			int myBias = W.depth16(last, gaps ? gaps - 1 : W.t_len - 1);
			const int tmp = W.depth16(0, gaps);
			myBias = (tmp < myBias) ? tmp : (myBias - 1);
			if(65535 < myBias) myBias = 65535;
			while(left > 0) {
				const int id = W.new_node(myBias, q_base(R.q, qpos++), gaps);
				if(!id) return;
				wg_fence();
				if(last) W.set_next(last, id); else W.set_head(gaps, id);
				last = id;
				--left;
			}
		} else for(; h; h = W.next(h)) W.add_node(h, 5);            // the rest of the chain: columns this read lacks
		// the column behind the insertion: first element of the next run (there is one: trailing gap runs are trimmed)
		const int ncls = j + 1 < R.n ? (int) (A.ops[R.o + j + 1] & 3u) : 0;
		W.add_col(gaps, ncls == 3 ? 5 : q_base(R.q, qpos));
	}
}


// the reads [s0, s1) of the sorted list -- one template's -- PILE_THREADS at a time
template <bool LDS>
__device__ void pile_template(const PileArgs &A, const Walk<LDS> &W, int64_t s0, int64_t s1, unsigned long long *s_ins) {
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	for(int64_t b = s0; b < s1; b += (int) blockDim.x) {
		const bool valid = b + tid < s1;
		const int64_t r = valid ? (int64_t) A.vals[b + tid] : 0;
		const unsigned long long m = __ballot(valid && read_has_ins(A, r));
		if(lane == 0) s_ins[wave] = m;
		__syncthreads();
		// a stretch of reads without an insertion run plus the aligned columns of the insertion-bearing read behind it go side
		// by side, G lanes a read; then that read's insertion runs, alone
		int cur = 0;
		while(cur < (int) blockDim.x) {
			int nxt = (int) blockDim.x;
			for(int w = cur >> 6; w < (int) blockDim.x / 64; ++w) {
				unsigned long long x = s_ins[w];
				if(w == (cur >> 6)) x &= ~0ull << (cur & 63);
				if(x) { nxt = (w << 6) + __ffsll((long long) x) - 1; break; }
			}
#ifdef KMAHIP_DIAG
			const unsigned long long c0 = wall_clock64();
#endif
			const int cnt = min(nxt + 1, (int) blockDim.x) - cur;
			int g_sh = 6;
			while(g_sh > 0 && (cnt << g_sh) > (int) blockDim.x) --g_sh;
			const int idx = cur + (tid >> g_sh);
			if(idx < cur + cnt && b + idx < s1) pile_cols(A, W, (int64_t) A.vals[b + idx], tid & ((1 << g_sh) - 1), 1 << g_sh);
			wg_fence();
			__syncthreads();
#ifdef KMAHIP_DIAG
			const unsigned long long c1 = wall_clock64();
#endif
			if(nxt < (int) blockDim.x) {
				if(tid == nxt) pile_ins(A, W, r);
				wg_fence();
				__syncthreads();
			}
#ifdef KMAHIP_DIAG
			if(tid == 0 && blockIdx.x == 7) { atomicAdd(&A.counters[10], c1 - c0); atomicAdd(&A.counters[11], wall_clock64() - c1); atomicAdd(&A.counters[12], 1ull); }
#endif
			cur = nxt + 1;
		}
		__syncthreads();
	}
}

__global__ __launch_bounds__(PILE_THREADS) void pileup_kernel(const PileArgs A, int64_t n_ent, int lds_cols) {
	__shared__ unsigned long long s_ins[PILE_THREADS / 64];
	__shared__ unsigned s_nodes;
	__shared__ int s_abort;
	__shared__ VisitDesc s_desc[3];
	__shared__ long long s_pool;
	const int tid = threadIdx.x;
	for(int64_t u = blockIdx.x; u < A.n_units; u += gridDim.x) {
		// a unit ran out of room somewhere: the host starts again with other sizes (one thread looks, so that all leave together)
		if(tid == 0) s_abort = ld_l2(&A.counters[1]) != 0 ? 1 : 0;
		__syncthreads();
		if(s_abort) break;
		const int64_t s0 = A.unit_start[u];
		if(s0 >= n_ent) continue;
		const int64_t s1 = A.unit_end[u];          // (the sorted list is unit-major)
		const int t = A.unit_t[u];
		const int64_t tbase = A.db.cat_off[t];
		const int t_len = A.db.tlen[t];
		if(A.unit_base[t + 1] - A.unit_base[t] > 1) {
			// one segment of a long template
			const int lo = A.unit_lo[u], hi = min(lo + A.seg_cols, t_len), ncol = hi - lo + 1;
			const SegWalk W{A, lo, hi, t_len, ncol, &s_nodes};
			for(int i = tid; i < 7 * ncol; i += (int) blockDim.x) pile_lds[i] = 0;
			if(tid == 0) s_nodes = 0;
			__syncthreads();
			// the visits in order. Three records in LDS: this visit's, the next one's (whose first runs every thread asks for now) and
			// the one after that (on its way from HBM)
			const uint32_t *dsrc = (const uint32_t *) (A.desc + s0);
			uint32_t *dlds = (uint32_t *) s_desc;
			if(tid < 32 && s0 + (tid >> 4) < s1) dlds[tid] = dsrc[tid];
			__syncthreads();
			uint32_t run0 = 0, prun0 = 0, nrun0 = 0;
			{
				const VisitDesc &V0 = s_desc[0];
				const int j = V0.j_begin + tid;
				if(j < V0.n) { run0 = A.ops[V0.o + j]; if(j > V0.first) prun0 = A.ops[V0.o + j - 1]; if(j + 1 < V0.n) nrun0 = A.ops[V0.o + j + 1]; }
			}
			for(int64_t e = s0; e < s1; ++e) {
				const int cur = (int) ((e - s0) % 3), nxt = (cur + 1) % 3, aft = (cur + 2) % 3;
				uint32_t dreg = 0, run1 = 0, prun1 = 0, nrun1 = 0;
				if(tid < 16 && e + 2 < s1) dreg = dsrc[(e + 2 - s0) * 16 + tid];
				if(e + 1 < s1) {
					const VisitDesc &V1 = s_desc[nxt];
					const int j = V1.j_begin + tid;
					if(j < V1.n) { run1 = A.ops[V1.o + j]; if(j > V1.first) prun1 = A.ops[V1.o + j - 1]; if(j + 1 < V1.n) nrun1 = A.ops[V1.o + j + 1]; }
				}
				pile_seg_read(A, W, s_desc[cur], run0, prun0, nrun0, s_ins);
				if(tid < 16 && e + 2 < s1) dlds[aft * 16 + tid] = dreg;
				if(tid == 0 && ((e - s0) & 63) == 63) s_abort = ld_l2(&A.counters[1]) != 0 ? 1 : 0;
				__syncthreads();
				if(s_abort) break;
				run0 = run1; prun0 = prun1; nrun0 = nrun1;
			}
			const int n_nodes = min((int) s_nodes, W.node_cap());
			if(tid == 0) s_pool = n_nodes ? (long long) atomicAdd(&A.counters[2], (unsigned long long) n_nodes) : 0;
			__syncthreads();
			const long long base = s_pool;
			if(base + n_nodes > A.node_cap) { if(tid == 0) atomicMax(&A.counters[1], 32ull); }
			else {
				for(int i = tid; i < 6 * (hi - lo); i += (int) blockDim.x) A.counts[6 * (tbase + lo) + i] = pile_lds[6 + i];
				for(int i = tid; i < hi - lo; i += (int) blockDim.x) { const int h = (int) pile_lds[6 * ncol + 1 + i]; A.chain_head[tbase + lo + i] = h ? (int32_t) (base + h) : 0; }
				for(int i = tid; i < n_nodes; i += (int) blockDim.x) {
					const uint32_t *c = &pile_lds[W.node_base() + 8 * i];
					InsNode &nn = A.nodes[base + i];
					for(int x = 0; x < 6; ++x) nn.c[x] = c[x];
					nn.next = c[6] ? (int32_t) (base + c[6]) : 0;
					nn.gaps = (int32_t) c[7];
				}
			}
			__syncthreads();
			continue;
		}
		if(t_len <= lds_cols) {
			const Walk<true> W{A, tbase, t_len, &s_nodes};
			for(int i = tid; i < 7 * t_len; i += (int) blockDim.x) pile_lds[i] = 0;
			if(tid == 0) s_nodes = 0;
			__syncthreads();
			pile_template(A, W, s0, s1, s_ins);
			// out to HBM; the insertion columns get a contiguous range of the pool (ids base + 1 ...)
			const int n_nodes = min((int) s_nodes, W.node_cap());
			if(tid == 0) s_pool = n_nodes ? (long long) atomicAdd(&A.counters[2], (unsigned long long) n_nodes) : 0;
			__syncthreads();
			const long long base = s_pool;
			if(base + n_nodes > A.node_cap) { if(tid == 0) atomicMax(&A.counters[1], 32ull); }
			else {
				for(int i = tid; i < 6 * t_len; i += (int) blockDim.x) A.counts[6 * tbase + i] = pile_lds[i];
				for(int i = tid; i < t_len; i += (int) blockDim.x) { const int h = (int) pile_lds[6 * t_len + i]; A.chain_head[tbase + i] = h ? (int32_t) (base + h) : 0; }
				for(int i = tid; i < n_nodes; i += (int) blockDim.x) {
					const uint32_t *c = &pile_lds[W.node_base() + 8 * i];
					InsNode &nn = A.nodes[base + i];
					for(int x = 0; x < 6; ++x) nn.c[x] = c[x];
					nn.next = c[6] ? (int32_t) (base + c[6]) : 0;
					nn.gaps = (int32_t) c[7];
				}
			}
			__syncthreads();
		} else pile_template(A, Walk<false>{A, tbase, t_len, nullptr}, s0, s1, s_ins);
	}
}


// ---- consensus on the device: callConsensus + baseCaller (assembly.c:1499-1631, 162-179) per column. The one piece of
// floating point in it, significantNuc's p_chisqr((X-Y)^2 / (X+Y)) <= evalue, is a monotone function of the quotient; the host
// finds the smallest double q* with p_chisqr(q*) <= evalue by bisection in its own libm arithmetic (and checks the
// neighbourhood for monotonicity), so the device only compares an IEEE double quotient with q*.
struct ConsArgs {
	DevDB db;
	const uint32_t *counts;
	const int32_t *chain_head;
	const InsNode *nodes;
	const int32_t *seg_start;
	int64_t n_kept;
	int bcd;
	int caller;                  // 0 baseCaller, 1 nanoCaller (-bcNano, assembly.c:205-240)
	int sig90;                   // 1: significantAnd90Nuc instead of significantNuc (-bcNano, -bc90; assembly.c:147-149), 2: significantAndSupport
	double support;
	int mark_ins;                // insertion columns called as gaps are written as '_' (kmahip_assemble_opts.caller bit 3)
	int mark_all;                // every insertion column's character carries bit 7 (caller bit 5: for the `.aln` writer)
	double qstar;
	unsigned long long *cover, *aln_len, *depth, *asm_len;      // per template
	char *cons;                  // pass 2: consensus characters
	const int64_t *cons_off;     // pass 2: per template offset into cons (-1: none)
	// the columns of a template are called in segments of CONS_SEG template positions, one workgroup each (a 5 Mb template by
	// one workgroup took 85 ms)
	const int32_t *cs_t, *cs_lo; // per segment: template, first position
	int64_t n_cs;
	unsigned long long *cs_items;   // pass 1 out: columns (template + insertion) of the segment
	const int64_t *cs_off;       // pass 2 in: columns of the template in front of the segment
};

__device__ __forceinline__ unsigned char dev_lower(unsigned char c) { return (c >= 'A' && c <= 'Z') ? (unsigned char) (c + 32) : c; }

__device__ unsigned char call_column_dev(const uint32_t *c32, int tnuc, int bcd, double qstar, int caller, int sig90, double support, long long *depth_out) {
	const char bases[7] = {'A', 'C', 'G', 'T', 'N', '-', 0};
	int cnt[6];
	for(int j = 0; j < 6; ++j) cnt[j] = (int) min(c32[j], 65535u);
	int bestNuc = tnuc;
	const char tch = bases[tnuc];
	int bestScore = cnt[bestNuc];
	long long depthUpdate = 0;
	for(int j = 0; j < 6; ++j) {
		if(bestScore < cnt[j]) { bestScore = cnt[j]; bestNuc = j; }
		depthUpdate += cnt[j];
	}
	unsigned char call = (unsigned char) bases[bestNuc];
	if(!depthUpdate) call = '-';
	else if(((long long) bestScore << 1) < depthUpdate) {
		if(call == '-') {
			int bb = cnt[4], b = 4;
			for(int j = 0; j < 4; ++j) if(bb < cnt[j]) { bb = cnt[j]; b = j; }
			call = dev_lower((unsigned char) bases[b]);
		} else call = dev_lower(call);
		bestScore = (int) (depthUpdate - cnt[5]);
	} else if(depthUpdate < bcd) call = dev_lower(call);
	{
		// the base callers (assembly.c:162-270): 0 baseCaller, 1 nanoCaller, 2 orgBaseCaller, 3 refCaller, 4 refNanoCaller
		const int X = bestScore, Y = (int) depthUpdate - bestScore;
		const bool sig = depthUpdate != 0 && Y < X && (sig90 != 1 || 9ll * (X + Y) <= 10ll * X) && (sig90 != 2 || support * (double) (X + Y) <= (double) X) &&
		                 ((double) ((long long) (X - Y) * (X - Y)) / (double) (X + Y)) >= qstar;
		auto best_base = [&](unsigned char none) {      // the best base count (N included) decides; first of equals
			int bb = 0, b = -1;
			for(int j = 0; j < 5; ++j) if(bb < cnt[j]) { bb = cnt[j]; b = j; }
			return bb == 0 ? none : dev_lower((unsigned char) bases[b]);
		};
		if(caller == 2) {
			if(depthUpdate == 0 || call == '-') call = '-';
			else if(!sig) call = dev_lower(call);
		} else if(caller == 3) {
			if(depthUpdate == 0 || (call == '-' && tch != '-')) call = 'n';
			else if(!sig) call = dev_lower(call);
		} else if(caller == 4) {
			if(depthUpdate == 0) call = 'n';
			else if(!sig) call = call == '-' ? best_base((unsigned char) 'n') : dev_lower(call);
			else if(call == '-') call = 'n';
		} else {
			if(depthUpdate == 0) call = '-';
			else if(!sig) {
				if(call == '-' && tch != '-' && bestScore != depthUpdate) call = caller == 1 ? best_base((unsigned char) '-') : (unsigned char) 'n';
				else call = dev_lower(call);
			}
		}
	}
	*depth_out = depthUpdate;
	return call;
}

constexpr int CONS_THREADS = 256;

constexpr int CONS_SEG = 8192;

// one workgroup per segment of a template with reads; ring order = template position p, then the insertion columns in front of p + 1
template <bool WRITE>
__global__ __launch_bounds__(CONS_THREADS) void consensus_kernel(const ConsArgs C) {
	__shared__ int s_scan[CONS_THREADS];
	__shared__ unsigned long long s_cover, s_aln, s_depth;
	const int tid = threadIdx.x;
	for(int64_t g = blockIdx.x; g < C.n_cs; g += gridDim.x) {
		const int t = C.cs_t[g];
		if(C.seg_start[t] >= C.n_kept) continue;           // uniform per workgroup
		const int t_len = C.db.tlen[t];
		const int lo = C.cs_lo[g], hi = min(lo + CONS_SEG, t_len);
		const int64_t base = C.db.cat_off[t];
		const uint64_t *ts = C.db.tseq + C.db.tseq_off[t];
		const int64_t coff = WRITE ? (C.cons_off[t] >= 0 ? C.cons_off[t] + C.cs_off[g] : -1) : -1;
		if(tid == 0) { s_cover = 0; s_aln = 0; s_depth = 0; }
		__syncthreads();
		unsigned long long cover = 0, aln = 0, depth = 0;
		int64_t running = 0;
		for(int b = lo; b < hi; b += CONS_THREADS) {
			const int p = b + tid;
			const bool valid = p < hi;
			int items = 0;
			if(valid) {
				const int np = (p + 1 == t_len) ? 0 : p + 1;
				items = 1;
				for(int h = C.chain_head[base + np]; h; h = C.nodes[h - 1].next) ++items;
			}
			// exclusive scan of `items` over the workgroup
			s_scan[tid] = items;
			__syncthreads();
			for(int o = 1; o < CONS_THREADS; o <<= 1) {
				const int v = tid >= o ? s_scan[tid - o] : 0;
				__syncthreads();
				s_scan[tid] += v;
				__syncthreads();
			}
			const int incl = s_scan[tid], tot = s_scan[CONS_THREADS - 1];
			__syncthreads();
			if(valid) {
				int64_t o = running + incl - items;
				const int tnuc = (int) ((ts[p >> 5] >> (62 - ((p & 31) << 1))) & 3ull);
				long long dep = 0;
				unsigned char call = call_column_dev(C.counts + (size_t) (base + p) * 6, tnuc, C.bcd, C.qstar, C.caller, C.sig90, C.support, &dep);
				if(WRITE && coff >= 0) C.cons[coff + o] = (char) call;
				++o;
				if(call != '-') {
					depth += (unsigned long long) dep; ++aln;
					const char up = (call >= 'a' && call <= 'z') ? (char) (call - 32) : (char) call;
					if("ACGTN-"[tnuc] == up) ++cover;
				}
				const int np = (p + 1 == t_len) ? 0 : p + 1;
				for(int h = C.chain_head[base + np]; h; h = C.nodes[h - 1].next) {
					call = call_column_dev(C.nodes[h - 1].c, 5, C.bcd, C.qstar, C.caller, C.sig90, C.support, &dep);
					// (an insertion column called as a gap is trimmed from the reference's alignment, assembly.c:748-752: marked where the
					// writer keeps the gaps of template positions, `-ref_fsa 0`)
					if(WRITE && coff >= 0) C.cons[coff + o] = (char) (((call == '-' && C.mark_ins) ? '_' : call) | (C.mark_all ? 0x80 : 0));
					++o;
					if(call != '-') { depth += (unsigned long long) dep; ++aln; }
				}
			}
			running += tot;
		}
		if(WRITE) {
			if(tid == 0 && coff >= 0 && hi == t_len) C.cons[coff + running] = 0;
		} else {
			atomicAdd(&s_cover, cover); atomicAdd(&s_aln, aln); atomicAdd(&s_depth, depth);
			__syncthreads();
			if(tid == 0) {
				atomicAdd(&C.cover[t], s_cover); atomicAdd(&C.aln_len[t], s_aln); atomicAdd(&C.depth[t], s_depth); atomicAdd(&C.asm_len[t], (unsigned long long) running);
				C.cs_items[g] = (unsigned long long) running;
			}
		}
		__syncthreads();
	}
}

} // namespace

namespace { struct DevGuard { std::vector<void *> v; ~DevGuard() { for(void *p : v) (void) hipFree(p); } }; }

static int assemble_scratch(kmahip_db *db, kmahip_ws *ws, int64_t n_reads, int64_t node_cap) {
	const int64_t total = db->h_cat_off.empty() ? 0 : db->h_cat_off.back();
	if(ws->p_total != total || ws->p_node_cap < node_cap) {
		(void) hipFree(ws->p_counts); (void) hipFree(ws->p_chain); (void) hipFree(ws->p_nodes); (void) hipFree(ws->p_seg);
		ws->p_counts = nullptr; ws->p_chain = nullptr; ws->p_nodes = nullptr; ws->p_seg = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->p_counts, (size_t) (total + 1) * 6 * sizeof(uint32_t)));
		HIP_TRY(hipMalloc((void **) &ws->p_chain, (size_t) (total + 1) * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->p_nodes, (size_t) node_cap * sizeof(InsNode)));
		HIP_TRY(hipMalloc((void **) &ws->p_seg, (size_t) (db->info.DB_size + 1) * sizeof(int32_t)));
		ws->p_total = total; ws->p_node_cap = node_cap;
	}
	if(ws->p_reads_cap < n_reads) {
		(void) hipFree(ws->p_rank);
		ws->p_rank = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->p_rank, (size_t) n_reads * 2 * sizeof(int64_t)));
		ws->p_reads_cap = n_reads;
	}
	if(!ws->counters) { HIP_TRY(hipMalloc((void **) &ws->counters, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); HIP_TRY(hipMemset(ws->counters, 0, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); }
	return KMAHIP_OK;
}

// device part: counts / chains of every template with kept reads. All pointers are device pointers.
// ---- alnToMatDense (`-dense`, assembly.c:1446-1497): template positions only -- the bases and gaps the reads put on them, no insertion
// columns -- so the order of the reads does not matter: a wavefront per read, its runs 64 at a time (prefix sums by shuffles), a lane
// per run. As the reference does it: gap runs at the END of an alignment are trimmed, those at its start are not (its loop for them
// never runs); a run of gaps in the template (an insertion) only moves on in the read.
__global__ __launch_bounds__(64) void pile_dense_kernel(const PileArgs A, unsigned long long *kept) {
	const int lane = threadIdx.x;
	for(int64_t r = blockIdx.x; r < A.n_reads; r += gridDim.x) {
		const int32_t *st = A.stats + 10 * r;
		if(st[3] == 0 || A.tmpl[r] == 0) continue;
		const int t = abs(A.tmpl[r]);
		const int t_len = A.db.tlen[t];
		const int64_t tbase = A.db.cat_off[t];
		if(lane == 0) { A.seg_start[t] = 0; atomicAdd(kept, 1ull); }
		const int64_t o = A.ops_off[r];
		int n = A.n_ops[r];
		while(n > 0 && (A.ops[o + n - 1] & 3u) >= 2u) --n;          // trailing gap runs
		Q q;
		q.w = A.seq + A.seq_off[r]; q.L = A.len[r]; q.N = A.N + A.N_off[r]; q.nN = (int) (A.N_off[r + 1] - A.N_off[r]);
		q.rc = (((A.flag[r] & 1) != 0) != (A.tmpl[r] < 0)) ? 1 : 0;
		q.cw = -1; q.cv = 0;
		int col_carry = 0, q_carry = st[4];
		const int start = st[1] >= t_len ? st[1] - t_len : st[1];
		for(int j0 = 0; j0 < n; j0 += 64) {
			const int j = j0 + lane;
			const bool valid = j < n;
			const uint32_t run = valid ? A.ops[o + j] : 0u;
			const int cls = (int) (run & 3u), len = (int) (run >> 2);
			const int tl = valid && cls != 2 ? len : 0, ql = valid && cls != 3 ? len : 0;
			int ct = tl, cq = ql;
			for(int d = 1; d < 64; d <<= 1) { const int a = __shfl_up(ct, d), b = __shfl_up(cq, d); if(lane >= d) { ct += a; cq += b; } }
			const int col0 = col_carry + ct - tl, q0 = q_carry + cq - ql;
			if(tl > 0) {
				int p = (start + col0) % t_len;
				for(int c = 0; c < len; ++c) {
					const int b = cls == 3 ? 5 : q_base(q, q0 + c);
					atomicAdd(&A.counts[6 * (tbase + p) + b], 1u);
					if(++p == t_len) p = 0;
				}
			}
			col_carry += __shfl(ct, 63); q_carry += __shfl(cq, 63);
		}
	}
}

static int pileup_dense_device(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl, const kmahip_traces *tr, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	int rc = assemble_scratch(db, ws, std::max<int64_t>(n, 1), 1 << 20);
	if(rc) return rc;
	const int64_t total = ws->p_total;
	const int64_t D = db->info.DB_size;
	PileArgs A;
	memset((void *) &A, 0, sizeof A);
	A.db = db->dev; A.n_reads = n; A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.flag = flag; A.tmpl = tmpl; A.stats = tr->stats; A.ops_off = tr->ops_off; A.n_ops = tr->n_ops; A.ops = tr->ops;
	A.counters = ws->counters; A.counts = ws->p_counts; A.chain_head = ws->p_chain; A.nodes = (InsNode *) ws->p_nodes; A.seg_start = ws->p_seg;
	ws->p_kept = 0; ws->p_nodes_used = 0;
	if(n == 0) return KMAHIP_OK;
	HIP_TRY(hipMemsetAsync(ws->p_counts, 0, (size_t) (total + 1) * 6 * sizeof(uint32_t), stream));
	HIP_TRY(hipMemsetAsync(ws->p_chain, 0, (size_t) (total + 1) * sizeof(int32_t), stream));
	HIP_TRY(hipMemsetAsync(ws->counters, 0, 3 * sizeof(unsigned long long), stream));
	{
		std::vector<int32_t> init((size_t) D + 1, INT32_MAX);
		HIP_TRY(hipMemcpyAsync(ws->p_seg, init.data(), init.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
		HIP_TRY(hipStreamSynchronize(stream));
	}
	hipLaunchKernelGGL(pile_dense_kernel, dim3((unsigned) std::min<int64_t>(n, 256 * 32)), dim3(64), 0, stream, A, ws->counters);
	HIP_TRY(hipGetLastError());
	unsigned long long kept = 0;
	HIP_TRY(hipMemcpy(&kept, ws->counters, 8, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemsetAsync(ws->counters, 0, 3 * sizeof(unsigned long long), stream));
	ws->p_kept = (int64_t) kept;
	return KMAHIP_OK;
}

static int pileup_device(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                         const kmahip_traces *tr, int64_t max_frag, int order, const int64_t *frag_rank, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	// insertion columns: about one per 30 read bases of noisy long reads at most (every site of a deep pile-up has a few)
	int64_t node_cap = std::max<int64_t>(1 << 20, n);
	if(reads->max_len > 1000) node_cap = std::max<int64_t>(node_cap, std::min<int64_t>((int64_t) reads->max_len * n / 8, 4 * (db->h_cat_off.empty() ? 0 : db->h_cat_off.back()) + (1 << 20)));
	int rc = assemble_scratch(db, ws, std::max<int64_t>(n, 1), node_cap);
	if(rc) return rc;
	const int64_t total = ws->p_total;
	const int64_t D = db->info.DB_size;
	PileArgs A;
	A.db = db->dev; A.n_reads = n; A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.flag = flag; A.tmpl = tmpl; A.stats = tr->stats; A.ops_off = tr->ops_off; A.n_ops = tr->n_ops; A.ops = tr->ops;
	A.max_frag = max_frag > 0 ? max_frag : 1000000;
	A.order = order;
	A.counters = ws->counters;
	A.counts = ws->p_counts; A.chain_head = ws->p_chain; A.nodes = (InsNode *) ws->p_nodes; A.node_cap = ws->p_node_cap; A.seg_start = ws->p_seg;
	A.lds_node_limit = getenv("KMAHIP_PILE_LDS_NODES") ? atoi(getenv("KMAHIP_PILE_LDS_NODES")) : 1 << 30;
	ws->p_kept = 0;
	if(n == 0) return KMAHIP_OK;
	if(n >= (1ll << 28)) { kmahip_set_error("pile-up: more than 2^28 reads in one batch"); return KMAHIP_EINVAL; }
	DevGuard G;
	void *tmp = nullptr;
	size_t tmp_bytes = 0;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto t_prev = now();
	auto lap = [&](const char *what) {
		if(!dbg) return;
		(void) hipStreamSynchronize(stream);
		const auto t = now();
		fprintf(stderr, "[kmahip] pile-up: %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
		t_prev = t;
	};
	if(frag_rank) A.rank = const_cast<int64_t *>(frag_rank);       // the batch is a gathered part of the stream: positions given
	else {
		// rank of every read among the filed fragments: exclusive scan of (tmpl != 0)
		int64_t *filed = ws->p_rank + n;
		A.rank = ws->p_rank;
		hipLaunchKernelGGL(pile_filed_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, tmpl, n, filed);
		if(rocprim::exclusive_scan(nullptr, tmp_bytes, filed, A.rank, (int64_t) 0, (size_t) n, rocprim::plus<int64_t>(), stream) != hipSuccess) {
			kmahip_set_error("rocprim::exclusive_scan (size query) failed"); return KMAHIP_EDEVICE;
		}
		HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
		const hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, filed, A.rank, (int64_t) 0, (size_t) n, rocprim::plus<int64_t>(), stream);
		HIP_TRY(hipStreamSynchronize(stream));
		(void) hipFree(tmp); tmp = nullptr;
		if(e != hipSuccess) { kmahip_set_error("rocprim::exclusive_scan failed: %s", hipGetErrorString(e)); return KMAHIP_EDEVICE; }
	}
	int64_t *vis = nullptr;
	HIP_TRY(hipMalloc((void **) &vis, (size_t) (2 * n + 2) * sizeof(int64_t)));
	G.v.push_back(vis);
	A.vis_cnt = vis; A.vis_off = vis + n + 1;
	// a template of up to split_cols columns is one unit (piled up in LDS as a whole, or on HBM if its insertion columns do not fit
	// there); a longer one is cut into segments of seg_cols columns, halved when a segment runs out of LDS room
	const int split_cols = (PILE_LDS_WORDS - 8 * PILE_LDS_MIN_NODES) / 7;
	int lds_cols = getenv("KMAHIP_PILE_NO_LDS") ? 0 : split_cols;
	int seg_cols = getenv("KMAHIP_PILE_SEG_COLS") ? atoi(getenv("KMAHIP_PILE_SEG_COLS")) : 1024;
	if(!getenv("KMAHIP_PILE_SEG_COLS")) {
		// a deep pile-up has a few insertion columns at every site, and they must fit beside the unit's columns in HALF the LDS (two
		// workgroups per CU are worth more than long units): shorter units from the start where the depth can be that large
		int64_t long_cols = 0;
		for(int64_t t = 1; t < D; ++t) if(db->h_tlen[(size_t) t] > split_cols) long_cols += db->h_tlen[(size_t) t];
		if(long_cols > 0) {
			const double depth = (double) n * (double) std::max(reads->max_len, 1) / (double) long_cols;
			while(seg_cols > 256 && depth * seg_cols > 1.2e6) seg_cols >>= 1;
		}
	}
	unsigned long long c[3] = {0, 0, 0};
	bool lds_half = !getenv("KMAHIP_PILE_FULL_LDS");
	for(;;) {
		std::vector<int32_t> unit_base((size_t) D + 1, 0), unit_t, unit_lo;
		for(int64_t t = 1; t < D; ++t) {
			unit_base[(size_t) t] = (int32_t) unit_t.size();
			const int tl = db->h_tlen[(size_t) t];
			if(tl <= split_cols) { unit_t.push_back((int32_t) t); unit_lo.push_back(0); }
			else for(int lo = 0; lo < tl; lo += seg_cols) { unit_t.push_back((int32_t) t); unit_lo.push_back(lo); }
		}
		unit_base[0] = 0; unit_base[(size_t) D] = (int32_t) unit_t.size();
		if(D > 1) unit_base[1] = 0;
		const int64_t n_units = (int64_t) unit_t.size();
		int32_t *d_units = nullptr;
		int64_t *d_ustart = nullptr;
		HIP_TRY(hipMalloc((void **) &d_units, (size_t) (D + 1 + 2 * n_units + 2) * sizeof(int32_t)));
		G.v.push_back(d_units);
		HIP_TRY(hipMalloc((void **) &d_ustart, (size_t) (2 * n_units + 4) * sizeof(int64_t)));
		G.v.push_back(d_ustart);
		HIP_TRY(hipMemcpyAsync(d_units, unit_base.data(), (size_t) (D + 1) * 4, hipMemcpyHostToDevice, stream));
		if(n_units) {
			HIP_TRY(hipMemcpyAsync(d_units + D + 1, unit_t.data(), (size_t) n_units * 4, hipMemcpyHostToDevice, stream));
			HIP_TRY(hipMemcpyAsync(d_units + D + 1 + n_units, unit_lo.data(), (size_t) n_units * 4, hipMemcpyHostToDevice, stream));
		}
		A.unit_base = d_units; A.unit_t = d_units + D + 1; A.unit_lo = d_units + D + 1 + n_units; A.n_units = n_units; A.seg_cols = seg_cols;
		A.unit_start = d_ustart; A.unit_end = d_ustart + n_units + 2;
		HIP_TRY(hipMemsetAsync(ws->p_counts, 0, (size_t) (total + 1) * 6 * sizeof(uint32_t), stream));
		HIP_TRY(hipMemsetAsync(ws->p_chain, 0, (size_t) (total + 1) * sizeof(int32_t), stream));
		HIP_TRY(hipMemsetAsync(ws->counters, 0, 3 * sizeof(unsigned long long), stream));
		{
			std::vector<int32_t> init((size_t) D + 1, INT32_MAX);
			HIP_TRY(hipMemcpyAsync(ws->p_seg, init.data(), init.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
			HIP_TRY(hipStreamSynchronize(stream));
		}
		// visits per read -> entries (unit, order) -> sorted
		hipLaunchKernelGGL(pile_count_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, A);
		tmp_bytes = 0;
		if(rocprim::exclusive_scan(nullptr, tmp_bytes, A.vis_cnt, A.vis_off, (int64_t) 0, (size_t) n + 1, rocprim::plus<int64_t>(), stream) != hipSuccess) {
			kmahip_set_error("rocprim::exclusive_scan (size query) failed"); return KMAHIP_EDEVICE;
		}
		HIP_TRY(hipMemsetAsync(A.vis_cnt + n, 0, sizeof(int64_t), stream));
		HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
		{
			const hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, A.vis_cnt, A.vis_off, (int64_t) 0, (size_t) n + 1, rocprim::plus<int64_t>(), stream);
			int64_t n_ent = 0;
			HIP_TRY(hipMemcpyAsync(&n_ent, A.vis_off + n, sizeof n_ent, hipMemcpyDeviceToHost, stream));
			HIP_TRY(hipStreamSynchronize(stream));
			(void) hipFree(tmp); tmp = nullptr;
			if(e != hipSuccess) { kmahip_set_error("rocprim::exclusive_scan failed: %s", hipGetErrorString(e)); return KMAHIP_EDEVICE; }
			ws->p_kept = n_ent;
		}
		const int64_t n_ent = ws->p_kept;
		if(!n_ent) return KMAHIP_OK;
		if(ws->p_ent_cap < n_ent) {
			(void) hipFree(ws->p_keys); (void) hipFree(ws->p_vals);
			ws->p_keys = nullptr; ws->p_vals = nullptr; ws->p_ent_cap = 0;
			HIP_TRY(hipMalloc((void **) &ws->p_keys, (size_t) n_ent * 2 * sizeof(uint64_t)));
			HIP_TRY(hipMalloc((void **) &ws->p_vals, (size_t) n_ent * 2 * sizeof(int32_t)));
			ws->p_ent_cap = n_ent;
		}
		A.keys = ws->p_keys; A.vals = ws->p_vals;
		hipLaunchKernelGGL(pile_keys_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, A);
		A.ck_j = nullptr; A.ck_col = nullptr; A.ck_q = nullptr;
		if(n_units > (int64_t) D - 1 && !getenv("KMAHIP_PILE_NO_CKPT")) {          // some template is cut into units
			int32_t *ck = nullptr;
			HIP_TRY(hipMalloc((void **) &ck, (size_t) n_ent * 3 * sizeof(int32_t)));
			G.v.push_back(ck);
			A.ck_j = ck; A.ck_col = ck + n_ent; A.ck_q = ck + 2 * n_ent;
			hipLaunchKernelGGL(pile_ckpt_kernel, dim3((unsigned) std::min<int64_t>(n, 256 * 32)), dim3(64), 0, stream, A);
		}
		lap("ranks + visits + keys");
		uint64_t *keys_out = ws->p_keys + n_ent;
		int32_t *vals_out = ws->p_vals + n_ent;
		tmp_bytes = 0;
		if(rocprim::radix_sort_pairs(nullptr, tmp_bytes, ws->p_keys, keys_out, ws->p_vals, vals_out, (size_t) n_ent, 0, 64, stream) != hipSuccess) {
			kmahip_set_error("rocprim::radix_sort_pairs (size query) failed"); return KMAHIP_EDEVICE;
		}
		HIP_TRY(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
		const hipError_t se = rocprim::radix_sort_pairs(tmp, tmp_bytes, ws->p_keys, keys_out, ws->p_vals, vals_out, (size_t) n_ent, 0, 64, stream);
		if(se != hipSuccess) { (void) hipFree(tmp); kmahip_set_error("rocprim::radix_sort_pairs failed: %s", hipGetErrorString(se)); return KMAHIP_EDEVICE; }
		A.keys = keys_out; A.vals = vals_out;
		hipLaunchKernelGGL(pile_fill_kernel, dim3((unsigned) ((n_units + 1 + 255) / 256)), dim3(256), 0, stream, d_ustart, n_units + 1, n_ent);
		hipLaunchKernelGGL(pile_segments_kernel, dim3((unsigned) ((n_ent + 255) / 256)), dim3(256), 0, stream, keys_out, n_ent, d_ustart, d_ustart + n_units + 2);
		A.desc = nullptr;
		if(n_units > (int64_t) D - 1) {
			VisitDesc *dd = nullptr;
			HIP_TRY(hipMalloc((void **) &dd, (size_t) (n_ent + 2) * sizeof(VisitDesc)));
			G.v.push_back(dd);
			A.desc = dd;
			hipLaunchKernelGGL(pile_desc_kernel, dim3((unsigned) ((n_ent + 255) / 256)), dim3(256), 0, stream, A, n_ent);
		}
		lap("sort + segments + visit records");
#ifdef KMAHIP_DIAG
		HIP_TRY(hipMemsetAsync(ws->counters + 10, 0, 6 * sizeof(unsigned long long), stream));
#endif
		const unsigned blocks = (unsigned) std::min<int64_t>(std::max<int64_t>(n_units, 1), 256 * 8);
		// A unit's time is a chain of dependent loads (its reads' runs, then the columns): with half the LDS each, two workgroups
		// share a CU and one's waiting hides behind the other's -- when every unit's columns fit that with room for a few hundred
		// insertion columns. (Status 16 below: a template needed more insertion columns -- again with all of the LDS, then on HBM.)
		int unit_cols = 0;
		for(int64_t t = 1; t < D; ++t) { const int tl = db->h_tlen[(size_t) t]; unit_cols = std::max(unit_cols, tl <= split_cols ? tl : std::min(seg_cols, tl) + 1); }
		const int half_words = (80 * 1024 - 1024) / 4;
		const bool half = lds_half && lds_cols && 7 * unit_cols + 8 * 512 <= half_words;
		A.lds_words = half ? half_words : PILE_LDS_WORDS;
		HIP_TRY(hipFuncSetAttribute((const void *) pileup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PILE_LDS_WORDS * 4));
		hipLaunchKernelGGL(pileup_kernel, dim3(blocks), dim3(half ? PILE_THREADS / 2 : PILE_THREADS), (size_t) A.lds_words * 4, stream, A, n_ent, lds_cols);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(stream));
		(void) hipFree(tmp); tmp = nullptr;
		lap("pileup_kernel");
		HIP_TRY(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost));
		if(c[1] == 64 && half && seg_cols > 256) { seg_cols >>= 1; if(dbg) fprintf(stderr, "[kmahip] pile-up: a segment ran out of its half of the LDS, again with %d columns per segment\n", seg_cols); continue; }
		if((c[1] == 16 || c[1] == 64) && half) { lds_half = false; if(dbg) fprintf(stderr, "[kmahip] pile-up: half the LDS did not hold a unit's insertion columns, again with all of it\n"); continue; }
		if(c[1] == 64 && seg_cols > 64) { seg_cols >>= 1; if(dbg) fprintf(stderr, "[kmahip] pile-up: a segment ran out of LDS room, again with %d columns per segment\n", seg_cols); continue; }
		if(c[1] == 16 && lds_cols) { lds_cols = 0; continue; }      // a template's insertion columns did not fit LDS: those on HBM
		break;
	}
	if(c[1]) {
		HIP_TRY(hipMemset(ws->counters + 1, 0, sizeof(unsigned long long)));
		kmahip_set_error("pile-up: insertion column pool exhausted (%lld columns, status %llu)", (long long) ws->p_node_cap, c[1]);
		return KMAHIP_EOVERFLOW;
	}
	ws->p_nodes_used = (int64_t) c[2];
#ifdef KMAHIP_DIAG
	{
		unsigned long long dbgc[6];
		HIP_TRY(hipMemcpy(dbgc, ws->counters + 10, sizeof dbgc, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemset(ws->counters + 10, 0, sizeof dbgc));
		fprintf(stderr, "[kmahip] pile-up workgroup 7: %llu phases, parallel part %.1f us, serial part %.1f us per phase; (100 MHz clock)\n", dbgc[2],
		        dbgc[2] ? dbgc[0] / 100.0 / dbgc[2] : 0.0, dbgc[2] ? dbgc[1] / 100.0 / dbgc[2] : 0.0);
	}
#endif
	return KMAHIP_OK;
}

// ---- consensus + `.res` columns, host arithmetic ----------------------------------------------------------------------
static double asm_chi2_table(long double q) {
	struct Step { double quantile, p; };
	static const Step steps[] = {
		{114.5242, 1e-26}, {109.9604, 1e-25}, {105.3969, 1e-24}, {100.8337, 1e-23}, {96.27476, 1e-22}, {91.71701, 1e-21},
		{87.16164, 1e-20}, {82.60901, 1e-19}, {78.05917, 1e-18}, {73.51245, 1e-17}, {68.96954, 1e-16}, {64.43048, 1e-15},
		{59.89615, 1e-14}, {55.36699, 1e-13}, {50.84417, 1e-12}, {46.32844, 1e-11}, {41.82144, 1e-10}, {37.32489, 1e-9},
		{32.84127, 1e-8}, {28.37395, 1e-7}, {23.92814, 1e-6}, {19.51139, 1e-5}, {15.13671, 1e-4}, {10.82759, 1e-3},
		{6.634897, 0.01}, {3.841443, 0.05}, {2.705532, 0.1}, {2.072251, 0.15}, {1.642374, 0.2}, {1.323304, 0.25},
		{1.074194, 0.3}, {0.8734571, 0.35}, {0.7083263, 0.4}, {0.5706519, 0.45}, {0.4549364, 0.5}, {0.3573172, 0.55},
		{0.2749959, 0.6}, {0.2059001, 0.65}, {0.1484719, 0.7}, {0.1015310, 0.75}, {0.06418475, 0.8}, {0.03576578, 0.85},
		{0.01579077, 0.9}, {0.00393214, 0.95} };
	for(const Step &st : steps) if(q > st.quantile) return st.p;
	if(q >= 0.0) return 1.0;
	return 1.00 - asm_chi2_table(-1 * q);
}
static double asm_p_chisqr(long double q) {      // stdstat.c:136-147
	if(q < 0) return 1e-26;
	if(q > 49) return asm_chi2_table(q);
	return 1 - 1.772453850 * erf(sqrt((double) (0.5 * q))) / tgamma(0.5);
}
static int significant_nuc(int X, int Y, double evalue, int sig90 = 0, double support = 0) {   // significantNuc / significantAnd90Nuc, assembly.c:143-149
	if(!(Y < X)) return 0;
	if(sig90 == 1 && !(9ll * (X + Y) <= 10ll * X)) return 0;
	if(sig90 == 2 && !(support * (double) (X + Y) <= (double) X)) return 0;
	// a pure function of (X, Y, evalue): memoised, most columns of a pile-up repeat a handful of (X, Y) pairs
	struct Slot { uint64_t key; double ev; int val; };
	static thread_local std::vector<Slot> memo(1 << 16, Slot{~0ull, 0.0, 0});
	const uint64_t key = ((uint64_t) (uint32_t) X << 32) | (uint32_t) Y;
	Slot &m = memo[(size_t) ((key * 0x9E3779B97F4A7C15ull) >> 48)];
	if(m.key == key && m.ev == evalue) return m.val;
	const int v = asm_p_chisqr(pow(X - Y, 2) / (X + Y)) <= evalue;
	m.key = key; m.ev = evalue; m.val = v;
	return v;
}

// one column: callConsensus body + baseCaller (assembly.c:1543-1595, 162-179); counts already clamped to 16 bit
static unsigned char call_column(const uint32_t *cnt, int tnuc /* 0-3 or 5 */, int bcd, double evalue, int caller, int sig90, double support, long *depth_out) {
	static const char bases[] = "ACGTN-";
	int bestNuc = tnuc;
	const char tch = bases[tnuc];
	int bestScore = (int) cnt[bestNuc];
	long depthUpdate = 0;
	for(int j = 0; j < 6; ++j) {
		if(bestScore < (int) cnt[j]) { bestScore = (int) cnt[j]; bestNuc = j; }
		depthUpdate += cnt[j];
	}
	unsigned char call = (unsigned char) bases[bestNuc];
	if(!depthUpdate) call = '-';
	else if(((long) bestScore << 1) < depthUpdate) {
		if(call == '-') {
			int bb = (int) cnt[4], b = 4;
			for(int j = 0; j < 4; ++j) if(bb < (int) cnt[j]) { bb = (int) cnt[j]; b = j; }
			call = (unsigned char) tolower(bases[b]);
		} else call = (unsigned char) tolower(call);
		bestScore = (int) (depthUpdate - cnt[5]);
	} else if(depthUpdate < bcd) call = (unsigned char) tolower(call);
	{
		const bool sig = depthUpdate != 0 && significant_nuc(bestScore, (int) depthUpdate - bestScore, evalue, sig90, support) != 0;
		auto best_base = [&](unsigned char none) {
			int bb = 0, b = -1;
			for(int j = 0; j < 5; ++j) if(bb < (int) cnt[j]) { bb = (int) cnt[j]; b = j; }
			return bb == 0 ? none : (unsigned char) tolower(bases[b]);
		};
		if(caller == 2) {
			if(depthUpdate == 0 || call == '-') call = '-';
			else if(!sig) call = (unsigned char) tolower(call);
		} else if(caller == 3) {
			if(depthUpdate == 0 || (call == '-' && tch != '-')) call = 'n';
			else if(!sig) call = (unsigned char) tolower(call);
		} else if(caller == 4) {
			if(depthUpdate == 0) call = 'n';
			else if(!sig) call = call == '-' ? best_base((unsigned char) 'n') : (unsigned char) tolower(call);
			else if(call == '-') call = 'n';
		} else {
			if(depthUpdate == 0) call = '-';
			else if(!sig) {
				if(call == '-' && tch != '-' && bestScore != depthUpdate) call = caller == 1 ? best_base((unsigned char) '-') : (unsigned char) 'n';
				else call = (unsigned char) tolower(call);
			}
		}
	}
	*depth_out = depthUpdate;
	return call;
}


extern "C" int kmahip_assemble(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                               const kmahip_traces *traces, int64_t max_frag, int bcd, double evalue, kmahip_assembly *out) {
	kmahip_assemble_opts o = {max_frag, evalue, bcd, 0, 0, 0};
	return kmahip_assemble2(db, ws, reads, flag, tmpl, traces, &o, out);
}

extern "C" int kmahip_assemble_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *d_flag, const int32_t *d_tmpl,
                                   const kmahip_traces *traces, int64_t max_frag, int bcd, double evalue, kmahip_assembly *out) {
	kmahip_assemble_opts o = {max_frag, evalue, bcd, 0, 0, 0};
	return kmahip_assemble2_dev(db, ws, reads, d_flag, d_tmpl, traces, &o, out);
}

extern "C" int kmahip_assemble2(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                                const kmahip_traces *traces, const kmahip_assemble_opts *opts, kmahip_assembly *out) {
	if(!opts) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(!db || !ws || !reads || !flag || !tmpl || !traces || !out || !out->cover || !out->aln_len || !out->depth || !out->asm_len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(db->h_cat_off.empty()) { kmahip_set_error("index has no .length.b/.seq.b: stage 3c unavailable"); return KMAHIP_EINVAL; }
	const int64_t D = db->info.DB_size;
	for(int64_t t = 0; t < D; ++t) { out->cover[t] = 0; out->aln_len[t] = 0; out->depth[t] = 0; out->asm_len[t] = 0; }
	if(n == 0) return KMAHIP_OK;
	int64_t total_ops = 0;
	for(int64_t i = 0; i < n; ++i) total_ops = std::max<int64_t>(total_ops, traces->ops_off[i] + traces->n_ops[i]);
	// stage everything (host buffers in): reads, flags, templates, traces
	std::vector<void *> owned;
	struct Free { std::vector<void *> &v; ~Free() { for(void *p : v) (void) hipFree(p); } } guard{owned};
	auto up = [&](const void *src, size_t bytes, void **dst) -> int {
		void *d = nullptr;
		if(hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) { kmahip_set_error("hipMalloc failed"); return KMAHIP_EDEVICE; }
		owned.push_back(d);
		if(bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
		*dst = d;
		return KMAHIP_OK;
	};
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	kmahip_traces dt = *traces;
	int32_t *d_flag, *d_tmpl;
	void *seq_d = nullptr;
	int rc;
	{
		std::vector<uint64_t> seq((size_t) reads->seq_words + 2, 0);
		if(reads->seq_words) memcpy(seq.data(), reads->seq, (size_t) reads->seq_words * 8);
		if((rc = up(seq.data(), seq.size() * 8, &seq_d))) return rc;
	}
	d.seq = (const uint64_t *) seq_d;
	if((rc = up(reads->seq_off, (size_t) (n + 1) * 8, (void **) &d.seq_off)) || (rc = up(reads->len, (size_t) n * 4, (void **) &d.len)) ||
	   (rc = up(reads->N, (size_t) reads->N_total * 4, (void **) &d.N)) || (rc = up(reads->N_off, (size_t) (n + 1) * 8, (void **) &d.N_off)) ||
	   (rc = up(flag, (size_t) n * 4, (void **) &d_flag)) || (rc = up(tmpl, (size_t) n * 4, (void **) &d_tmpl)) ||
	   (rc = up(traces->stats, (size_t) n * 40, (void **) &dt.stats)) || (rc = up(traces->ops_off, (size_t) n * 8, (void **) &dt.ops_off)) ||
	   (rc = up(traces->n_ops, (size_t) n * 4, (void **) &dt.n_ops)) || (rc = up(traces->ops, (size_t) total_ops * 4, (void **) &dt.ops))) return rc;
	kmahip_assemble_opts od = *opts;
	if(opts->frag_rank && (rc = up(opts->frag_rank, (size_t) n * 8, (void **) &od.frag_rank))) return rc;
	return kmahip_assemble2_dev(db, ws, &d, d_flag, d_tmpl, &dt, &od, out);
}

// the same with the per-read inputs already in HBM (reads, rc, tmpl, traces: DEVICE pointers; `out`: host)
extern "C" int kmahip_assemble2_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *d_flag, const int32_t *d_tmpl,
                                    const kmahip_traces *traces, const kmahip_assemble_opts *opts, kmahip_assembly *out) {
	if(!opts) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t max_frag = opts->max_frag;
	const int bcd = opts->bcd, caller = opts->caller & 7, sig90 = opts->sig90, mark_ins = (opts->caller >> 3) & 1, mark_all = (opts->caller >> 5) & 1;
	const double support = opts->support;
	const double evalue = opts->evalue;
	if(!db || !ws || !reads || !d_flag || !d_tmpl || !traces || !out || !out->cover || !out->aln_len || !out->depth || !out->asm_len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(db->h_cat_off.empty()) { kmahip_set_error("index has no .length.b/.seq.b: stage 3c unavailable"); return KMAHIP_EINVAL; }
	const int64_t D = db->info.DB_size;
	for(int64_t t = 0; t < D; ++t) { out->cover[t] = 0; out->aln_len[t] = 0; out->depth[t] = 0; out->asm_len[t] = 0; }
	if(reads->n_reads == 0) return KMAHIP_OK;
	int rc;
	const kmahip_reads &d = *reads;
	const kmahip_traces &dt = *traces;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
	const auto t0 = now();
	if((opts->caller & 16) ? (rc = pileup_dense_device(db, ws, &d, d_flag, d_tmpl, &dt, 0)) : (rc = pileup_device(db, ws, &d, d_flag, d_tmpl, &dt, max_frag, opts->order, opts->frag_rank, 0))) return rc;
	const auto t1 = now();
	if(dbg) fprintf(stderr, "[kmahip] assemble: pile-up on device %.1f ms\n", ms(t0, t1));
	if(!ws->p_kept) return KMAHIP_OK;

	// consensus on the device (two passes: figures + lengths, then the characters at their offsets); the host version below is
	// kept as the arithmetic reference -- KMAHIP_HOST_CONSENSUS=1 selects it, and it takes over if p_chisqr is not monotone
	// around the threshold in this libm
	if(!getenv("KMAHIP_HOST_CONSENSUS")) {
		double qstar = 0.0;
		bool usable = true;
		if(asm_p_chisqr(0.0L) <= evalue) qstar = 0.0;
		else {
			double lo = 0.0, hi = 200.0;                       // p(lo) > evalue >= p(hi)
			if(!(asm_p_chisqr(hi) <= evalue)) usable = false;    // evalue below 1e-26: nothing is ever significant
			while(usable) {
				uint64_t a, b;
				memcpy(&a, &lo, 8); memcpy(&b, &hi, 8);
				if(b - a <= 1) break;
				const uint64_t m = a + (b - a) / 2;
				double mid;
				memcpy(&mid, &m, 8);
				if(asm_p_chisqr(mid) <= evalue) hi = mid; else lo = mid;
			}
			qstar = hi;
			// monotone around the threshold? (a few thousand neighbouring doubles on each side)
			uint64_t qb;
			memcpy(&qb, &qstar, 8);
			for(int i = 1; usable && i <= 4096; ++i) {
				double above, below;
				const uint64_t ua = qb + (uint64_t) i, ub = qb - (uint64_t) i;
				memcpy(&above, &ua, 8); memcpy(&below, &ub, 8);
				if(!(asm_p_chisqr(above) <= evalue) || (asm_p_chisqr(below) <= evalue)) usable = false;
			}
		}
		if(usable) {
			DevGuard G;
			ConsArgs C;
			C.db = db->dev; C.counts = ws->p_counts; C.chain_head = ws->p_chain; C.nodes = (const InsNode *) ws->p_nodes; C.seg_start = ws->p_seg;
			C.n_kept = ws->p_kept; C.bcd = bcd; C.caller = caller; C.sig90 = sig90; C.support = support; C.mark_ins = mark_ins; C.mark_all = mark_all; C.qstar = qstar; C.cons = nullptr; C.cons_off = nullptr;
			unsigned long long *fig = nullptr;
			HIP_TRY(hipMalloc((void **) &fig, (size_t) 4 * D * sizeof(unsigned long long)));
			G.v.push_back(fig);
			HIP_TRY(hipMemset(fig, 0, (size_t) 4 * D * sizeof(unsigned long long)));
			C.cover = fig; C.aln_len = fig + D; C.depth = fig + 2 * D; C.asm_len = fig + 3 * D;
			std::vector<int32_t> cs_t, cs_lo;
			std::vector<int64_t> cs_first((size_t) D + 1, 0);
			for(int64_t t = 1; t < D; ++t) {
				cs_first[(size_t) t] = (int64_t) cs_t.size();
				for(int lo = 0; lo < db->h_tlen[(size_t) t]; lo += CONS_SEG) { cs_t.push_back((int32_t) t); cs_lo.push_back(lo); }
			}
			cs_first[(size_t) D] = (int64_t) cs_t.size();
			const int64_t n_cs = (int64_t) cs_t.size();
			int32_t *d_cs = nullptr;
			unsigned long long *d_items = nullptr;
			HIP_TRY(hipMalloc((void **) &d_cs, (size_t) (2 * n_cs + 2) * 4)); G.v.push_back(d_cs);
			HIP_TRY(hipMalloc((void **) &d_items, (size_t) (2 * n_cs + 2) * 8)); G.v.push_back(d_items);
			HIP_TRY(hipMemset(d_items, 0, (size_t) (2 * n_cs + 2) * 8));
			if(n_cs) {
				HIP_TRY(hipMemcpy(d_cs, cs_t.data(), (size_t) n_cs * 4, hipMemcpyHostToDevice));
				HIP_TRY(hipMemcpy(d_cs + n_cs, cs_lo.data(), (size_t) n_cs * 4, hipMemcpyHostToDevice));
			}
			C.cs_t = d_cs; C.cs_lo = d_cs + n_cs; C.n_cs = n_cs; C.cs_items = d_items; C.cs_off = (const int64_t *) (d_items + n_cs + 1);
			const unsigned blocks = (unsigned) std::min<int64_t>(std::max<int64_t>(n_cs, 1), 256 * 16);
			hipLaunchKernelGGL((consensus_kernel<false>), dim3(blocks), dim3(CONS_THREADS), 0, 0, C);
			std::vector<unsigned long long> hf((size_t) 4 * D);
			HIP_TRY(hipMemcpy(hf.data(), fig, hf.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
			for(int64_t t = 1; t < D; ++t) {
				out->cover[t] = (int64_t) hf[(size_t) t]; out->aln_len[t] = (int64_t) hf[(size_t) (D + t)];
				out->depth[t] = (int64_t) hf[(size_t) (2 * D + t)]; out->asm_len[t] = (int64_t) hf[(size_t) (3 * D + t)];
			}
			if(out->consensus && out->consensus_off) {
				std::vector<int64_t> coff((size_t) D, -1);
				int64_t used = 0;
				for(int64_t t = 1; t < D; ++t) if(out->asm_len[t] > 0) { coff[(size_t) t] = used; used += out->asm_len[t] + 1; }
				if(out->consensus_used + used > out->consensus_cap) { kmahip_set_error("consensus_cap too small: %lld bytes needed, %lld given", (long long) (out->consensus_used + used), (long long) out->consensus_cap); return KMAHIP_EOVERFLOW; }
				char *dc = nullptr;
				int64_t *dco = nullptr;
				HIP_TRY(hipMalloc((void **) &dc, (size_t) used + 16)); G.v.push_back(dc);
				HIP_TRY(hipMalloc((void **) &dco, (size_t) D * 8)); G.v.push_back(dco);
				HIP_TRY(hipMemcpy(dco, coff.data(), (size_t) D * 8, hipMemcpyHostToDevice));
				{
					// columns in front of every segment inside its template
					std::vector<unsigned long long> items((size_t) n_cs + 1, 0);
					std::vector<int64_t> soff((size_t) n_cs + 1, 0);
					if(n_cs) HIP_TRY(hipMemcpy(items.data(), d_items, (size_t) n_cs * 8, hipMemcpyDeviceToHost));
					for(int64_t t = 1; t < D; ++t) {
						int64_t run = 0;
						for(int64_t g = cs_first[(size_t) t]; g < cs_first[(size_t) t + 1]; ++g) { soff[(size_t) g] = run; run += (int64_t) items[(size_t) g]; }
					}
					if(n_cs) HIP_TRY(hipMemcpy(d_items + n_cs + 1, soff.data(), (size_t) n_cs * 8, hipMemcpyHostToDevice));
				}
				C.cons = dc; C.cons_off = dco;
				hipLaunchKernelGGL((consensus_kernel<true>), dim3(blocks), dim3(CONS_THREADS), 0, 0, C);
				HIP_TRY(hipMemcpy(out->consensus + out->consensus_used, dc, (size_t) used, hipMemcpyDeviceToHost));
				for(int64_t t = 1; t < D; ++t) if(coff[(size_t) t] >= 0) out->consensus_off[t] = out->consensus_used + coff[(size_t) t];
				out->consensus_used += used;
			}
			HIP_TRY(hipGetLastError());
			if(dbg) fprintf(stderr, "[kmahip] assemble: consensus on device %.1f ms (q* = %.17g)\n", ms(t1, now()), qstar);
			return KMAHIP_OK;
		}
	}
	// consensus on the host
	const int64_t total = ws->p_total;
	std::vector<uint32_t> counts((size_t) total * 6);
	std::vector<int32_t> chain((size_t) total);
	std::vector<InsNode> nodes((size_t) ws->p_nodes_used);
	std::vector<int32_t> seg((size_t) D + 1);
	HIP_TRY(hipMemcpy(counts.data(), ws->p_counts, counts.size() * 4, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(chain.data(), ws->p_chain, chain.size() * 4, hipMemcpyDeviceToHost));
	if(!nodes.empty()) HIP_TRY(hipMemcpy(nodes.data(), ws->p_nodes, nodes.size() * sizeof(InsNode), hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(seg.data(), ws->p_seg, seg.size() * 4, hipMemcpyDeviceToHost));
	std::vector<uint64_t> tseq((size_t) db->info.tseq_words + 2);
	HIP_TRY(hipMemcpy(tseq.data(), db->dev.tseq, (size_t) db->info.tseq_words * 8, hipMemcpyDeviceToHost));
	std::vector<int64_t> toff((size_t) D + 1);
	HIP_TRY(hipMemcpy(toff.data(), db->dev.tseq_off, toff.size() * 8, hipMemcpyDeviceToHost));
	std::string cons;
	const auto t2 = now();
	if(dbg) fprintf(stderr, "[kmahip] assemble: copy back %.1f ms\n", ms(t1, t2));
	for(int64_t t = 1; t < D; ++t) {
		if(seg[t] >= ws->p_kept) continue;              // no read was piled up on this template
		const int t_len = db->h_tlen[t];
		const int64_t base = db->h_cat_off[t];
		const uint64_t *ts = tseq.data() + toff[t];
		int64_t cover = 0, aln_len = 0, depth = 0, asm_len = 0;
		cons.clear();
		auto column = [&](const uint32_t *c32, int tnuc, bool is_template) {
			uint32_t c[6];
			for(int j = 0; j < 6; ++j) c[j] = std::min<uint32_t>(c32[j], 65535u);
			long dep = 0;
			const unsigned char call = call_column(c, tnuc, bcd, evalue, caller, sig90, support, &dep);
			++asm_len;
			if(out->consensus) cons.push_back((char) (((call == '-' && !is_template && mark_ins) ? '_' : call) | (!is_template && mark_all ? 0x80 : 0)));
			if(call != '-') {
				depth += dep; ++aln_len;
				if(is_template && "ACGTN-"[tnuc] == toupper(call)) ++cover;
			}
		};
		for(int p = 0; p < t_len; ++p) {
			// ring order: template position p, then the insertion columns in front of position p + 1
			column(counts.data() + (size_t) (base + p) * 6, (int) ((ts[p >> 5] >> (62 - ((p & 31) << 1))) & 3ull), true);
			const int np = (p + 1 == t_len) ? 0 : p + 1;
			for(int h = chain[(size_t) (base + np)]; h; h = nodes[(size_t) h - 1].next) column(nodes[(size_t) h - 1].c, 5, false);
		}
		out->cover[t] = cover; out->aln_len[t] = aln_len; out->depth[t] = depth; out->asm_len[t] = asm_len;
		if(out->consensus && out->consensus_off) {
			if((int64_t) (out->consensus_used + cons.size() + 1) > out->consensus_cap) { kmahip_set_error("consensus_cap too small: more than %lld bytes needed", (long long) out->consensus_cap); return KMAHIP_EOVERFLOW; }
			out->consensus_off[t] = out->consensus_used;
			memcpy(out->consensus + out->consensus_used, cons.data(), cons.size());
			out->consensus[out->consensus_used + cons.size()] = 0;
			out->consensus_used += (int64_t) cons.size() + 1;
		}
	}
	if(dbg) fprintf(stderr, "[kmahip] assemble: consensus on host %.1f ms\n", ms(t2, now()));
	return KMAHIP_OK;
}

// the `.res` row exactly as runKMA prints it (runkma.c:792-809); returns 0 when the reference prints no row
// (no covered position, or identity / depth below the -ID / -md thresholds), else the number of characters written
extern "C" int kmahip_res_line(const char *template_name, const kmahip_res_row *row, int64_t cover, int64_t aln_len, int64_t depth_sum,
                               double ID_t, double Depth_t, char *line, int64_t cap) {
	if(!template_name || !row || !line || cap <= 0) return 0;
	if(!(cover > 0)) return 0;
	const int t_len = row->template_length;
	long double depth = depth_sum;
	depth /= t_len;
	const double id = 100.0 * cover / t_len;
	const double q_id = 100.0 * cover / aln_len;
	const double cov = 100.0 * aln_len / t_len;
	const double q_cover = 100.0 * t_len / aln_len;
	if(!(ID_t <= id && 0 < id && Depth_t <= depth)) return 0;
	// runkma.c:141: expected / q_value are long double and printed as (unsigned) / (double)
	const int w = snprintf(line, (size_t) cap, "%s\t%8ld\t%8u\t%8d\t%8.2f\t%8.2f\t%8.2f\t%8.2f\t%8.2f\t%8.2f\t%4.1e\n", template_name,
	                       (long) row->score, row->expected, t_len, id, cov, q_id, q_cover, (double) depth, row->q_value, row->p_value);
	return (w > 0 && w < cap) ? w : 0;
}
