// longtrace.hip -- the traceback aligner as a pipeline of kernels with work compaction in between, built for reads that
// carry hundreds of MEMs (BASELINE config C4: 10 kb Nanopore-style reads, `-Mt1 t -bcNano`), where one lane per read with
// its move matrix in HBM (trace_kernel, align.hip) is structurally wrong. Behaviour restated from
//   anker_rc     align.c:780-991    both strands of a raw read seeded against the template, strand by MEM coverage
//   KMA          align.c:214-507    chain, leading / trailing tails, joins with traceback
//   chainSeeds   chain.c:79-260     NW nw.c:26-309    NW_band nw.c:310-640
//   assemble_KMA assembly.c:1917-1965 (read filter)
//
//   lt_seed_kernel    one WAVEFRONT per read: k-mer lookups 256 positions at a time (4 gathers per lane in flight), maximal
//                     matches of all hits side by side, the reference's sequential walk over them out of LDS, chainSeeds with
//                     the lanes over the 127 successors, the chain's joins -> one DP problem descriptor per join, appended
//                     to a queue per size class
//   lt_dp_kernel<W>   W = 8 / 16 / 32 / 64 lanes per problem: anti-diagonal sweep, lane n owns query column n, the move
//                     matrix E (one byte per cell, the reference's encoding + a mismatch bit) in LDS, then one lane per
//                     problem walks E and writes the alignment columns as runs
//   lt_dpx_kernel     problems of 65..255 query columns and banded problems (several columns per lane, E in LDS or HBM);
//                     whatever fits neither is solved by one lane with rows and E in HBM (same code path as the reference)
//   lt_finish_kernel  per read: the runs of its problems and MEMs concatenated in chain order, read filter, output
#include "kmahip_internal.h"
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "dna_dev.h"
#include <algorithm>
#include <vector>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <unistd.h>

namespace {

#ifndef KMAHIP_LT_TILE
#define KMAHIP_LT_TILE 256
#endif
constexpr int LT_TILE = KMAHIP_LT_TILE;   // query positions looked up per round of the seeding wavefront
constexpr int LT_NJ = LT_TILE / 64;   // ... per lane
constexpr int LT_NEXT_CAP = 512;     // MEMs of the winning strand whose chain links / chain order stay in LDS
constexpr int LT_NCLS = 17;           // problem classes: 0..3 = 8 / 16 / 32 / 64 lanes per problem; 4, 5 = full matrix of up to 128 / 255
                                      // columns, 6, 7 = banded of up to 128 / 255 columns (several columns per lane); 8 = the rest (one
                                      // lane); 9..12 = 4..7 with a move matrix too large for LDS (kept in the workgroup's HBM scratch);
                                      // 15, 16 = banded matrix of up to 511 / 1023 columns (8 / 16 per lane, HBM): the tails of a gene
                                      // found in the middle of a long read (13, 14: their full-matrix counterparts, not instantiated)
constexpr int LT_E_WAVE = 8192;       // bytes of move matrix per wavefront in lt_dp_kernel (split between its problems)
constexpr int LT_TMAX = 127;          // template rows of a problem in lt_dp_kernel
constexpr int LT_XE_LDS = 32768;      // bytes of move matrix in LDS per workgroup of lt_dpx_kernel
constexpr int LT_XT_LDS = 2048;       // template rows staged in LDS there

enum { PF_NONE = 1, PF_DEGEN_I = 2, PF_DEGEN_D = 4, PF_LEAD_TRIM = 8, PF_TRAIL_TRIM = 16 };
// counters of a longtrace pass: [1] status, [2] problem pool top, [3] run pool top (words), [4..20] class counts,
// [21] DP cells, [22] MEMs of the chained strands (work figures), [23] problems in the lane queue, [24..] of them per lane class,
// [LC_OUT] output run pool top
constexpr int LT_LFULL = 6, LT_LCLS = 9;   // lane classes (lt_lane_kernel / lt_lane_band_kernel): 0..5 full matrix with rows of up to
                                      // 16 / 32 / 48 / 64 / 128 / 256 cells, 6..8 banded with rows of up to 72 / 96 / 144 (lt_lane_geom)
enum { LC_STATUS = 1, LC_PROB = 2, LC_RUNS = 3, LC_CNT = 4, LC_CELLS = 21, LC_MEMS = 22, LC_LANE = 23, LC_LCNT = 24, LC_OUT = 24 + LT_LCLS, LC_N = 25 + LT_LCLS };
// (a counter per 128-byte line: the seeding's wavefronts take their pool places and queue places with returning atomics, a million and a
// half of them per pass, and atomics on one line are served one after the other)
constexpr int LCS = 16;
#define LCI(x) ((x) * LCS)

struct LtRead {               // per read of the pass
	int64_t first;            // first problem descriptor
	int32_t status;           // 0: no alignment (unmapped, chain below the thresholds, join too large), 1: aligned
	int32_t rc;               // 1: the reverse complement of the read is what was aligned
	int32_t n_prob;           // chain length + 1 problem slots: leading tail, joins, trailing tail
	int32_t mapQ;
	int32_t pos0;             // template position in front of the first MEM (Stat.pos before the leading tail)
	int32_t clip0;            // query bases in front of the leading tail's problem
	int32_t qe_trail;         // query end of the trailing tail's problem
	int32_t pad;
};

struct LtProb {               // one join / tail: 64 bytes
	int64_t runs;             // word offset of its run slot in the run pool
	int32_t read;             // read of the pass
	int32_t t_s, t_l;         // template start (0-based) and rows (may wrap around a circular template)
	int32_t q_s, q_l;         // oriented query start and columns
	int32_t k;                // mode: 0 global, -1 / -2 free leading template / both, 1 / 2 free trailing
	int32_t band;             // 0: full matrix, else NW_band with this band
	int32_t flags;            // PF_*
	int32_t body;             // '=' columns of the MEM that follows the problem
	int32_t body_score;
	// results
	int32_t score;
	int32_t n_runs;
	int32_t clip;             // leading tail: query bases the walk left in front; trailing tail: behind
	int32_t pad;
};

struct LtArgs {
	DevDB db;
	int64_t r0, n_reads;      // reads [r0, r0 + n_reads) of the batch form this pass
	const uint64_t *seq;
	const int64_t *seq_off;
	const int32_t *len;
	const int32_t *N;
	const int64_t *N_off;
	const int32_t *tmpl;      // per read (batch index): signed template, or NULL: tmpl_all
	const int32_t *rc_in;     // per read: orientation to align (NULL: both strands are seeded, anker_rc decides)
	int score_mode;           // 1: KMA_score (align.c:509-748) instead of KMA(): its seeding rule at the end of a stretch, raw figures out, no runs
	const int32_t *q_start, *q_end;   // query bounds per read (records of the default mode; with rc_in only), NULL: whole reads
	const uint8_t *tmpl_ok;   // per template: align its reads? (NULL: all)
	int tmpl_all;
	int one2one, exhaustive;
	int M, MM, U, W1, Wl;
	int d[25];
	int minlen, mq;
	int ts;                   // -ts (trimSeeds, chain.c:493-528): not in score mode (KMA_score has no such step)
	double scoreT, mrc;
	// scratch of the pass
	int32_t *mem;             // per seeding wavefront: 7 arrays of mcap ints (tS tE qS qE weight next chain)
	int mcap;
	LtRead *rd;
	LtProb *prob;
	int64_t prob_cap;
	uint32_t *runs;           // run slots of the problems
	int64_t runs_cap;
	int32_t *queue;           // LT_NCLS x prob_cap problem indices
	int32_t *lq;              // the lane queue: problem indices of all lane classes (prob_cap), sorted by lkey before the kernels run
	uint32_t *lkey;           // (lane class << 24) | iterations of the problem's sweep
	int stop;                 // diagnosis (KMAHIP_LT_STOP): 1 = reads end after their seeding, 2 = after the chain
	int lane_turns;           // a lane takes problems of up to this many cells (one per turn, a quarter of a microsecond each: the longest problem of
	                          // a class is what its kernel lasts; longer ones go to the wavefront-per-problem kernels). KMAHIP_LT_LANE_TURNS
	int lane_mask;            // diagnosis (KMAHIP_LT_LANE=f / b): 1 = full-matrix classes only, 2 = banded only, 3 = both
	int lane_tq;              // 0: no lane classes; else the largest rows + columns whose scores stay inside 16 bits
	uint32_t *tmp;            // per finishing wavefront: tmp_cap words
	int64_t tmp_cap;
	uint8_t *xE;              // per lt_dpx workgroup: xe_cap bytes of move matrix + 4 rows of xrow ints
	int64_t xe_cap;
	int xrow;
	unsigned long long *counters;
	// output (kmahip_traces of the whole batch) + strand
	int32_t *o_stats;
	int64_t *o_off;
	int32_t *o_nops;
	uint32_t *ops;
	int64_t ops_cap;
	unsigned long long *ops_top;   // the batch's run pool top (shared by the passes)
	int32_t *o_rc;
	volatile uint32_t *rec;   // bring-up flight recorder in host memory (KMAHIP_DEBUG_TIMING): 16 words per workgroup, NULL = off
};
#define LT_REC(slot, val) do { if(A.rec && blockIdx.x < 2048) A.rec[(size_t) blockIdx.x * 16 + (slot)] = (uint32_t) (val); } while(0)

__device__ __forceinline__ int wave_max(int x) {
	for(int o = 32; o > 0; o >>= 1) x = max(x, __shfl_xor(x, o));
	return x;
}
__device__ __forceinline__ int wave_sum(int x) {
	for(int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
	return x;
}
__device__ __forceinline__ int wave_scan_incl(int x, int lane) {
	for(int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(x, d); if(lane >= d) x += v; }
	return x;
}
__device__ __forceinline__ void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
}
// the same when what one lane wrote to HBM is read by the others. Workgroup scope is enough: a workgroup of these kernels is one
// wavefront, its lanes share the CU's L1, which a store writes through and updates; the agent scope this was written with first
// writes back and invalidates L2 on every call (buffer_wbl2 / buffer_inv on gfx950) -- per read and per problem that was the
// larger part of the seeding kernel's time
__device__ __forceinline__ void wave_sync_hbm() {
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	__builtin_amdgcn_wave_barrier();
}

// position-index lookup of one template, table pointer and shift in registers
__device__ __forceinline__ int lt_lookup(const uint2 *tab, uint32_t sh, uint32_t km) {
	if(km == 0) return 0;
	const uint32_t msk = (1u << (32 - sh)) - 1u;
	uint32_t sl = (km * 0x9E3779B1u) >> sh;
	for(;;) {
		const uint2 e = tab[sl];
		if(e.y == 0) return 0;
		if(e.x == km) return (int) e.y;
		sl = (sl + 1u) & msk;
	}
}

// exact match lengths around the k-mer hit (query position s, template position pos1, 1-based): fwd = matching bases from s on
// (>= k; bounded by the N-free stretch [lowq, segE) and the template end), bwd = matching bases in front of s
__device__ __forceinline__ void lt_extend(const uint64_t *ts, int t_len, const QView &q, int s, int pos1, int k, int lowq, int segE, int &fwd, int &bwd) {
	int kk = s - 1, prev = pos1 - 2;
	for(;;) {
		const int room = min(kk - lowq + 1, prev + 1);
		if(room <= 0) break;
		const int step = min(32, room);
		const uint64_t xq = qwin(q, kk - step + 1) >> (64 - 2 * step);
		const uint64_t xt = win2(ts, prev - step + 1) >> (64 - 2 * step);
		const uint64_t x = xq ^ xt;
		const int same = x ? (__ffsll((long long) x) - 1) >> 1 : step;
		kk -= same; prev -= same;
		if(same < step) break;
	}
	bwd = s - 1 - kk;
	int value = pos1 + k - 1, l = s + k;
	for(;;) {
		const int room = min(segE - l, t_len - value);
		if(room <= 0) break;
		const int step = min(32, room);
		const uint64_t x = (qwin(q, l) ^ win2(ts, value)) >> (64 - 2 * step);
		const int same = x ? (__clzll((long long) x) - (64 - 2 * step)) >> 1 : step;
		l += same; value += same;
		if(same < step) break;
	}
	fwd = l - s;
}

struct SeedLds {
	// (the seeding's tile and the chain's window are never in use together: one piece of LDS for both, 6.5 kB a wavefront instead of 9.5)
	union {
		struct { int v[LT_TILE], F[LT_TILE], B[LT_TILE]; };
		struct { int ring[6][128]; int stage[5][64]; };
	};
	uint16_t next[LT_NEXT_CAP], chain[LT_NEXT_CAP];
	int d[25];
};

// MEM arrays of a seeding wavefront in HBM
struct MemArr {
	int32_t *tS, *tE, *qS, *qE, *w, *nx, *ch;
};

// The seeding loop of anker_rc / KMA() for one orientation of the read (align.c:812-945): every k-mer start of an N-free
// stretch is looked up -- a stretch is (re)entered only while MORE than k bases remain in it (`i < end - kmersize`), then
// scanned up to its last k-mer; a unique hit gives one MEM and the scan resumes behind it, a duplicated k-mer one MEM per
// occurrence and the scan resumes behind the longest. The lookups and the match lengths of all hits of 256 positions are
// done side by side; the walk itself is the same for every lane (LDS broadcast reads), lane 0 writes the MEMs.
// Returns false when the MEM arrays are full.
__device__ bool lt_seed_strand(const LtArgs &A, SeedLds &S, const MemArr &Mm, const QView &q, const uint64_t *ts, int t_len,
                               const uint2 *tab, uint32_t tsh, int k, int &tot, int &mem_count, int &score_r) {
	const int lane = threadIdx.x & 63;
	const int q_len = q.L;
	// query bounds (KMA(), align.c:249-270): the scan starts at q_start; a stretch ends at the next N, the last one at q_end; a MEM
	// is extended backwards to the N before it (or the read's first base), whatever q_start says
	const int q_stop = qb1(q);
	// (where a scan starts or starts again -- at the head of a stretch, behind a MEM -- KMA() wants more than k bases in front of it,
	// align.c:256, 306, 367; KMA_score is content with k, align.c:541)
	const int kk = A.score_mode ? k - 1 : k;
	int segS = q.b0, ni = 1;
	while(segS < q_stop) {
		while(ni <= q.nN && qN_at(q, ni) < segS) ++ni;
		const int segE = ni <= q.nN ? qN_at(q, ni) : q_stop;      // next N (oriented)
		const int lowq = ni > 1 ? qN_at(q, ni - 1) + 1 : 0;
		int cur = segS;
		bool scanning = false;
		int carry_pos = -2, carry_v = 0, carry_F = 0, carry_B = 0;
		for(int guard = 0;; ++guard) {
			if(guard > q_len + 2) { if(lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 11ull); return false; }     // (cur advances every round)
			if(!scanning) { if(!(cur < segE - kk)) break; scanning = true; }
			if(cur > segE - k) break;
			const int p0 = cur;
			// ---- lookups of the k-mer starts p0 .. p0 + 255 (position p0 + j * 64 + lane) ----
			int v[LT_NJ];
			uint32_t km[LT_NJ];
#pragma unroll
			for(int j = 0; j < LT_NJ; ++j) {
				const int pos = p0 + j * 64 + lane;
				km[j] = (pos <= segE - k) ? (uint32_t) (qwin(q, pos) >> (64 - 2 * k)) : 0u;
			}
			{
				// first probes of the four lookups travel together; a probe takes the aligned pair of slots its slot lies in (16 bytes:
				// the chain of a lookup is walked two slots per round trip, and with the table at most a third full (db.hip) the
				// longest of a round's 256 chains is two or three round trips instead of fifteen)
				const uint32_t msk = (1u << (32 - tsh)) - 1u;
				uint32_t sl[LT_NJ];
				uint4 e[LT_NJ];
				// A k-mer the database's presence bits (db.hip: 2 MiB at most, L2-resident) do not know is in no template's index: its
				// lookup goes to the table's first slots with every other such lookup -- one line for all of them -- and counts as a
				// miss. Against one genome nine lookups in ten miss, and the lookups run at two thirds of the box's gather ceiling.
				bool maybe[LT_NJ];
				if(A.db.kbits) {
					uint32_t wd[LT_NJ];
#pragma unroll
					for(int j = 0; j < LT_NJ; ++j) wd[j] = A.db.kbits[((km[j] * KMAHIP_KBITS_MUL) >> A.db.kbits_shift) >> 5];
#pragma unroll
					for(int j = 0; j < LT_NJ; ++j) maybe[j] = km[j] != 0 && ((wd[j] >> (((km[j] * KMAHIP_KBITS_MUL) >> A.db.kbits_shift) & 31u)) & 1u);
				} else {
#pragma unroll
					for(int j = 0; j < LT_NJ; ++j) maybe[j] = km[j] != 0;
				}
#pragma unroll
				for(int j = 0; j < LT_NJ; ++j) { sl[j] = maybe[j] ? (km[j] * 0x9E3779B1u) >> tsh : 0u; e[j] = *(const uint4 *) (tab + (sl[j] & ~1u)); }
#pragma unroll
				for(int j = 0; j < LT_NJ; ++j) {
					v[j] = 0;
					if(maybe[j]) for(;;) {
						if(!(sl[j] & 1u)) {
							if(e[j].y == 0) break;
							if(e[j].x == km[j]) { v[j] = (int) e[j].y; break; }
						}
						if(e[j].w == 0) break;
						if(e[j].z == km[j]) { v[j] = (int) e[j].w; break; }
						sl[j] = ((sl[j] | 1u) + 1u) & msk; e[j] = *(const uint4 *) (tab + sl[j]);
					}
				}
			}
			// ---- runs of hits on one diagonal: the first of a run extends, the others derive from it ----
			int hd[LT_NJ];
			unsigned long long hitm[LT_NJ];
			int prev_last = (carry_pos == p0 - 1) ? carry_v : 0;
#pragma unroll
			for(int j = 0; j < LT_NJ; ++j) {
				int pv = __shfl_up(v[j], 1);
				if(lane == 0) pv = prev_last;
				prev_last = __shfl(v[j], 63);
				const int idx = j * 64 + lane;
				const bool cont = v[j] > 0 && pv > 0 && pv + 1 == v[j];
				int F = 0, B = 0;
				hd[j] = -1;
				if(v[j] > 0 && !cont) {
					lt_extend(ts, t_len, q, p0 + idx, v[j], k, lowq, segE, F, B);
					hd[j] = idx;
				} else if(cont && idx == 0) {
					F = carry_F - 1; B = carry_B + 1;       // continues the last run of the round before
					hd[j] = 0;
				}
				S.v[idx] = v[j];
				if(hd[j] >= 0) { S.F[idx] = F; S.B[idx] = B; }
				hitm[j] = __ballot(v[j] != 0);
			}
			wave_sync();
			{
				int carry = -1;
#pragma unroll
				for(int j = 0; j < LT_NJ; ++j) {
					int x = hd[j];
					for(int dd = 1; dd < 64; dd <<= 1) { const int y = __shfl_up(x, dd); if(lane >= dd) x = max(x, y); }
					x = max(x, carry);
					carry = __shfl(x, 63);
					const int idx = j * 64 + lane;
					if(v[j] > 0 && hd[j] < 0 && x >= 0) { S.F[idx] = S.F[x] - (idx - x); S.B[idx] = S.B[x] + (idx - x); }
				}
			}
			wave_sync();
			carry_pos = p0 + LT_TILE - 1; carry_v = S.v[LT_TILE - 1]; carry_F = S.F[LT_TILE - 1]; carry_B = S.B[LT_TILE - 1];
			if(carry_v <= 0) carry_pos = -2;
			// ---- the walk over this round's hits ----
			for(;;) {
				// first hit at or behind cur
				const int off = cur - p0;
				int s = -1;
#pragma unroll
				for(int j = 0; j < LT_NJ; ++j) {
					if(s >= 0) continue;
					unsigned long long m = hitm[j];
					const int lo = off - j * 64;
					if(lo >= 64) continue;
					if(lo > 0) m &= ~0ull << lo;
					if(m) s = j * 64 + __ffsll((long long) m) - 1;
				}
				if(s < 0) { cur = p0 + LT_TILE; break; }         // none left: the scan goes on in the next round
				const int val = S.v[s];
				const int qs = p0 + s;
				if(val > 0) {
					if(tot >= A.mcap) return false;
					const int F = S.F[s], B = S.B[s];
					if(lane == 0) { Mm.qS[tot] = qs - B; Mm.tS[tot] = val - B; Mm.qE[tot] = qs + F; Mm.tE[tot] = val + F; Mm.w[tot] = F + B; }
					score_r += F + B;
					++tot; ++mem_count;
					cur = qs + F;
				} else {
					// duplicated k-mer: one MEM per occurrence (ascending positions), on behind the longest
					const int32_t *dl = A.db.tpos_dups + (-val - 1);
					const int cnt = dl[0];
					int bias = qs;
					for(int c = 1; c <= cnt; ++c) {
						if(tot >= A.mcap) return false;
						int F, B;
						const int pos1 = dl[c];
						lt_extend(ts, t_len, q, qs, pos1, k, lowq, segE, F, B);
						if(lane == 0) { Mm.qS[tot] = qs - B; Mm.tS[tot] = pos1 - B; Mm.qE[tot] = qs + F; Mm.tE[tot] = pos1 + F; Mm.w[tot] = F + B; }
						++tot; ++mem_count;
						bias = max(bias, qs + F);
					}
					score_r += k + (bias - qs);
					cur = bias + 1;
				}
				scanning = false;
				if(cur <= qs) { if(lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 11ull); return false; }
				if(!(cur < segE - kk)) break;
				scanning = true;
				if(cur >= p0 + LT_TILE) break;
			}
			if(!scanning) break;
		}
		segS = segE + 1;
	}
	return true;
}

// chainSeeds (chain.c:79-260) over the MEMs [base, base + n) of the wavefront: right to left, the 127 successors of a MEM
// two per lane out of an LDS ring, the reference's left-to-right acceptance rule (`<=` for a free join, `<` for the
// overlapping ones) resolved from the wave maximum: the later of the free joins that reach it, else the first join that does.
// Returns the best start (relative to base); links in S.next (n <= LT_NEXT_CAP) or Mm.nx.
__device__ int lt_chain(const LtArgs &A, SeedLds &S, const MemArr &Mm, int base, int n, int q_len, int t_len, int k, unsigned *mapQ, int *bestScore) {
	const int lane = threadIdx.x & 63;
	const int W1 = A.W1, U = A.U, M = A.M, MM = A.MM;
	const bool lds_next = n <= LT_NEXT_CAP;
	int best = 0, second = 0, bestPos = n - 1, bestW = 0;
	for(int i = n - 1; i >= 0; --i) {
		if((i & 63) == 63 || i == n - 1) {
			// stage the MEMs of this block of 64
			wave_sync();
			const int b0 = i & ~63, m = b0 + lane;
			if(m < n) {
				S.stage[0][lane] = Mm.tS[base + m]; S.stage[1][lane] = Mm.tE[base + m]; S.stage[2][lane] = Mm.qS[base + m];
				S.stage[3][lane] = Mm.qE[base + m]; S.stage[4][lane] = Mm.w[base + m];
			}
			wave_sync();
		}
		const int si = i & 63;
		const int tSi = S.stage[0][si], tEnd = S.stage[1][si], qSi = S.stage[2][si], qEnd = S.stage[3][si], wi = S.stage[4][si];
		const int weight = wi * M;
		int span = min(t_len - tEnd, q_len - qEnd);
		int gap = span - 1;
		gap = gap ? gap * U + W1 : W1;
		int sub = mism_score(span, k, M, MM);
		const int score0 = weight + (sub < gap ? gap : sub);
		// candidates
		const int lim = min(n, i + 128);
		int g[2];
		bool ok[2], free_join[2];
#pragma unroll
		for(int h = 0; h < 2; ++h) {
			const int j = i + 1 + h * 64 + lane;
			ok[h] = false; free_join[h] = false; g[h] = INT_MIN;
			if(j < lim) {
				const int sl = j & 127;
				const int tSj = S.ring[0][sl], tEj = S.ring[1][sl], qSj = S.ring[2][sl], qEj = S.ring[3][sl], scj = S.ring[4][sl];
				if(qEnd < qSj) {
					if(tEnd < tSj) {
						const int tGap = tSj - tEnd, qGap = qSj - qEnd;
						int x = abs(tGap - qGap);
						if(x) x = (x - 1) * U + W1;
						g[h] = x + weight + scj + mism_score(min(tGap, qGap), k, M, MM);
						ok[h] = true; free_join[h] = true;
					} else if(k <= tEj - tEnd) {
						int x = qSj - qEnd;
						if(x) x = (x - 1) * U + W1;
						g[h] = x + weight + scj - (tSj - tEnd) * M;
						ok[h] = true;
					}
				} else if(k <= qEj - qEnd) {
					const int tStart = tSj + qEnd - qSj;
					if(tEnd < tStart) {
						int x = tStart - tEnd;
						if(x) x = (x - 1) * U + W1;
						g[h] = x + weight + scj - (tStart - tEnd) * M;
						ok[h] = true;
					}
				}
			}
		}
		const int G = wave_max(max(g[0], g[1]));
		int nxt = 0, score = score0;
		if(G >= score0) {
			const unsigned long long e0 = __ballot(ok[0] && g[0] == G), e1 = __ballot(ok[1] && g[1] == G);
			const unsigned long long a0 = __ballot(free_join[0] && g[0] == G), a1 = __ballot(free_join[1] && g[1] == G);
			int lastA = -1;
			if(a1) lastA = i + 65 + (63 - __clzll((long long) a1));
			else if(a0) lastA = i + 1 + (63 - __clzll((long long) a0));
			if(G > score0) {
				const int first = e0 ? i + 1 + (__ffsll((long long) e0) - 1) : i + 65 + (__ffsll((long long) e1) - 1);
				nxt = lastA >= 0 ? lastA : first;
				score = G;
			} else if(lastA >= 0) nxt = lastA;
		}
		const int wnew = nxt ? (wi + S.ring[5][nxt & 127] - k + 1) : (wi - (k - 1));
		wave_sync();
		if(lane == 0) {
			const int sl = i & 127;
			S.ring[0][sl] = tSi; S.ring[1][sl] = tEnd; S.ring[2][sl] = qSi; S.ring[3][sl] = qEnd; S.ring[4][sl] = score; S.ring[5][sl] = wnew;
			if(lds_next) S.next[i] = (uint16_t) nxt; else Mm.nx[base + i] = nxt;
		}
		wave_sync();
		span = min(tSi, qSi);
		gap = span - 1;
		if(0 < gap) gap = gap * U + W1; else if(gap == 0) gap = W1; else gap = 0;
		sub = mism_score(span, k, M, MM);
		score += sub < gap ? gap : sub;
		if(best <= score) {
			if(nxt != bestPos) second = best;
			best = score; bestPos = i; bestW = wnew;
		} else if(second <= score && nxt != bestPos) {
			second = best;
		}
	}
	if(0 < best) {
		const double wq = fmin(1.0, bestW / 10.0);
		*mapQ = (unsigned) ceil(40 * (1 - 1.0 * second / best) * wq * log((double) best));
	} else *mapQ = 0;
	*bestScore = best;
	return bestPos;
}

// one join of the chain (KMA(), align.c:430-466): from the end of MEM a to the start of MEM b, which gives way where the
// two overlap. l == 0: leading tail in front of b (leadTailAln, align.c:53-131); l == nc: trailing tail behind a
// (trailTailAln, align.c:140-212).
struct Join { int t_s, t_l, q_s, q_l, k, band, flags, body, bq, fail, clip0, qe; };
__device__ Join lt_join(const LtArgs &A, const MemArr &Mm, int ia, int ib, int l, int nc, int q_len, int t_len) {
	const int bw = 64;
	Join J;
	J.t_s = 0; J.t_l = 0; J.q_s = 0; J.q_l = 0; J.k = 0; J.band = 0; J.flags = PF_NONE; J.body = 0; J.bq = 0; J.fail = 0; J.clip0 = 0; J.qe = 0;
	if(l == 0) {
		const int t_e = Mm.tS[ib] - 1, q_e = Mm.qS[ib];
		J.body = Mm.qE[ib] - Mm.qS[ib]; J.bq = q_e;
		if(q_e) {
			int t_s = 0, q_s = 0;
			if((q_e << 1) < t_e || (q_e + bw) < t_e) t_s = t_e - (q_e + (q_e < bw ? q_e : bw));
			else if((t_e << 1) < q_e || (t_e + bw) < q_e) q_s = q_e - (t_e + (t_e < bw ? t_e : bw));
			J.clip0 = q_s;
			if(t_e - t_s > 0 && q_e - q_s > 0) {
				const int band = abs(t_e - t_s - q_e + q_s) + bw;
				J.t_s = t_s; J.t_l = t_e - t_s; J.q_s = q_s; J.q_l = q_e - q_s; J.k = -1 - (t_s == 0);
				J.band = (q_e - q_s <= band || t_e - t_s <= band) ? 0 : band;
				J.flags = (t_s == 0 && !A.score_mode) ? PF_LEAD_TRIM : 0;          // (KMA_score keeps the gap columns at the template's ends: no Frag_align, align.c:95, 174)
			}
		}
		return J;
	}
	if(l == nc) {
		const int t_s = Mm.tE[ia] - 1, q_s = Mm.qE[ia];
		int q_e = q_len, t_e = t_len;
		if(((q_len - q_s) << 1) < (t_len - t_s) || (q_len - q_s + bw) < (t_len - t_s)) {
			t_e = q_len - q_s; t_e = t_s + (t_e + (t_e < bw ? t_e : bw));
		} else if(((t_len - t_s) << 1) < (q_len - q_s) || (t_len - t_s + bw) < (q_len - q_s)) {
			q_e = t_len - t_s; q_e = q_s + (q_e + (q_e < bw ? q_e : bw));
		}
		J.qe = q_e;
		if(t_e - t_s > 0 && q_e - q_s > 0) {
			const int band = abs(t_e - t_s - q_e + q_s) + bw;
			J.t_s = t_s; J.t_l = t_e - t_s; J.q_s = q_s; J.q_l = q_e - q_s; J.k = 1 + (t_e == t_len);
			J.band = (q_e - q_s <= band || t_e - t_s <= band) ? 0 : band;
			J.flags = (t_e == t_len && !A.score_mode) ? PF_TRAIL_TRIM : 0;
		}
		return J;
	}
	const int q_s = Mm.qE[ia], t_s = Mm.tE[ia] - 1;
	int qSn = Mm.qS[ib], tSn = Mm.tS[ib];
	const int tEb = Mm.tE[ib], qEb = Mm.qE[ib];
	if(qSn < q_s) { tSn += q_s - qSn; qSn = q_s; }
	int t_e = tSn - 1, t_l;
	if(t_e < t_s) {
		if(t_s <= tEb) { qSn += t_s - t_e; t_e = t_s; t_l = 0; }
		else t_l = t_len - t_s + t_e;
	} else t_l = t_e - t_s;
	const int q_e = qSn;
	J.body = qEb - qSn; J.bq = qSn;
	if(abs(t_l - q_e + q_s) * A.U > q_len * A.M || t_l > q_len || q_e - q_s > (q_len >> 1)) { J.fail = 1; return J; }
	if(t_l > 0 || q_e - q_s > 0) {
		J.t_s = t_s; J.t_l = t_l; J.q_s = q_s; J.q_l = q_e - q_s; J.k = 0;
		if(t_l == 0) J.flags = PF_DEGEN_I;
		else if(q_e - q_s == 0) J.flags = PF_DEGEN_D;
		else {
			const int band = abs(t_l - q_e + q_s) + bw;
			J.band = (q_e - q_s <= band || t_l <= band) ? 0 : band;
			J.flags = 0;
		}
	}
	return J;
}

// size class of a DP problem
__device__ __forceinline__ int lt_class(int q_l, int t_l, int band, int k, int64_t xe_cap) {
	if(band == 0 && t_l <= LT_TMAX) {
		const int cells = (q_l + 1) * (t_l + 1);
		if(q_l <= 8 && cells <= LT_E_WAVE / 8) return 0;
		if(q_l <= 16 && cells <= LT_E_WAVE / 4) return 1;
		if(q_l <= 32 && cells <= LT_E_WAVE / 2) return 2;
		if(q_l <= 64 && cells <= LT_E_WAVE) return 3;
	}
	if(band & 1) ++band;
	const int pitch = band ? band + 2 : q_l + 1;
	// the reference's last-row scan (k == -2) reads cells of its row buffer beyond the boundary column when the band is cut
	// there: only the one-lane form reproduces that
	const int cfin = ((t_l + q_l) >> 1) - (t_l - 1);
	const bool stale_scan = band && k == -2 && !(cfin + (band >> 1) < q_l - 1);
	if(q_l < 256 && !stale_scan && (int64_t) pitch * (t_l + 1) <= xe_cap)
		return ((int64_t) pitch * (t_l + 1) <= LT_XE_LDS ? 4 : 9) + (band ? 2 : 0) + (q_l <= 128 ? 0 : 1);
	// (banded only: the full-matrix instantiations of 8 and 16 columns per lane spill scalar registers by the dozen)
	if(band && q_l < 1024 && !stale_scan && (int64_t) pitch * (t_l + 1) <= xe_cap) return 15 + (q_l < 512 ? 0 : 1);
	return 8;
}

// lane class of a DP problem (lt_lane_kernel: one LANE per problem, rows in LDS as 16-bit pairs, move matrix in HBM), -1: none.
// Geometry of class j: rows of up to R = 16 << j cells (full matrix: q_l + 1 of them; banded: band + 3), a move matrix of up to
// 4096 << j bytes, up to 255 (j <= 2) / 511 template rows. *iters = turns of the problem's sweep.
struct LaneGeom { int R, RQ, TW, ecap; };   // cells per row, query columns, template words (16 rows each), bytes of move matrix
__host__ __device__ __forceinline__ LaneGeom lt_lane_geom(int j) {
	// LDS per wavefront = (32 + 64 R + 64 TW) x 4 + 64 RQ (full matrix) or + 32 RQ (banded: two columns per byte) bytes, and a CU has
	// 160 KB: 16 / 11 / 8 / 6 / 3 / 1 wavefronts per CU for the full-matrix classes, 6 / 4 / 2 for the banded ones -- the sweeps wait
	// for LDS and for their own instructions' results, so the wavefronts per CU are what they run at
	LaneGeom g;
	if(j < LT_LFULL) {
		const int R[LT_LFULL] = {16, 32, 48, 64, 128, 256};
		g.R = R[j]; g.RQ = g.R; g.TW = j < 4 ? 16 : 32; g.ecap = g.R * 256;
	}
	else if(j == LT_LFULL) { g.R = 72; g.RQ = 144; g.TW = 12; g.ecap = 72 * 192; }
	else if(j == LT_LFULL + 1) { g.R = 96; g.RQ = 192; g.TW = 16; g.ecap = 96 * 256; }
	else { g.R = 144; g.RQ = 256; g.TW = 32; g.ecap = 65536; }
	return g;
}
__device__ __forceinline__ int lt_lane_class(int q_l, int t_l, int band, int k, int tq_max, int mask, int turn_cap, int *iters) {
	if(!tq_max || t_l + q_l > tq_max || t_l < 1 || q_l < 1) return -1;
	if(!(mask & (band ? 2 : 1))) return -1;
	if(band == 0) {
		const int64_t e = (int64_t) (q_l + 1) * (t_l + 1) + 4;
		for(int j = 0; j < LT_LFULL; ++j) {
			const LaneGeom g = lt_lane_geom(j);
			if(!((mask >> (8 + j)) & 1)) continue;
			if(q_l + 1 <= g.R && e <= g.ecap && t_l < 16 * g.TW) { *iters = (q_l + 1) * t_l; return *iters <= turn_cap ? j : -1; }
		}
		return -1;
	}
	if(band & 1) ++band;
	// (the reference's last-row scan over cells the band no longer covers: see lt_class)
	const int cfin = ((t_l + q_l) >> 1) - (t_l - 1);
	if(k == -2 && !(cfin + (band >> 1) < q_l - 1)) return -1;
	const int64_t e = (int64_t) (band + 2) * (t_l + 1) + 4;
	for(int j = LT_LFULL; j < LT_LCLS; ++j) {
		const LaneGeom g = lt_lane_geom(j);
		if(!((mask >> (8 + j)) & 1)) continue;
		if(band + 3 <= g.R && q_l <= g.RQ && e <= g.ecap && t_l < 16 * g.TW) { *iters = (band + 2) * t_l; return *iters <= turn_cap ? j : -1; }
	}
	return -1;
}

// (five wavefronts per SIMD: 94 registers, three spilled dwords; with the tile and the chain's window sharing one piece of LDS twenty
// wavefronts fit a CU and leave the sweeps that run beside them room -- the stage at 200 k reads 174 -> 161 ms with the LDS alone, 155 with both)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void lt_seed_kernel(const LtArgs A) {
	__shared__ SeedLds S;
	const int lane = threadIdx.x;
	if(lane < 25) S.d[lane] = A.d[lane];
	const int k = (int) A.db.kmersize;
	MemArr Mm;
	{
		int32_t *m0 = A.mem + (size_t) blockIdx.x * 7 * A.mcap;
		Mm.tS = m0; Mm.tE = m0 + A.mcap; Mm.qS = m0 + 2 * (size_t) A.mcap; Mm.qE = m0 + 3 * (size_t) A.mcap; Mm.w = m0 + 4 * (size_t) A.mcap;
		Mm.nx = m0 + 5 * (size_t) A.mcap; Mm.ch = m0 + 6 * (size_t) A.mcap;
	}
	// reads are dealt out round robin (a work counter with `if(lane == 0) atomicAdd` + broadcast in these kernels was turned by
	// the compiler into loops that drop lane 0 from EXEC on gfx950 -- plain strides are scalar loops and cannot go wrong that way)
	for(int64_t rr = blockIdx.x; rr < A.n_reads; rr += gridDim.x) {
		const int64_t r = A.r0 + rr;
		LtRead H;
		H.first = 0; H.status = 0; H.rc = 0; H.n_prob = 0; H.mapQ = 0; H.pos0 = 0; H.clip0 = 0; H.qe_trail = 0; H.pad = 0;
		const int tt = A.tmpl ? A.tmpl[r] : A.tmpl_all;
		const int t = abs(tt);
		QView qf;
		qf.w = A.seq + A.seq_off[r]; qf.L = A.len[r]; qf.N = A.N + A.N_off[r]; qf.nN = (int) (A.N_off[r + 1] - A.N_off[r]); qf.rc = 0;
		const int q_len = qf.L;
		bool go = t != 0 && !(A.tmpl_ok && !A.tmpl_ok[t]) && q_len > 0;
		int t_len = 0, base = 0, n = 0;
		const uint64_t *ts = nullptr;
		if(go) {
			const uint4 ma = A.db.tmeta[2 * (size_t) t], mb = A.db.tmeta[2 * (size_t) t + 1];
			t_len = (int) mb.x;
			ts = A.db.tseq + (((uint64_t) ma.y << 32) | ma.x);
			const uint2 *tab = A.db.tpos_slots + (((uint64_t) ma.w << 32) | ma.z);
			const uint32_t tsh = mb.y;
			int tot = 0;
			bool room = true;
			if(A.rc_in) {
				// orientation known (the read ConClave filed under a template): KMA()'s own seeding, one strand
				qf.rc = (((A.rc_in[r] & 1) != 0) != (tt < 0)) ? 1 : 0;
				q_set_bounds(qf, A.q_start, A.q_end, r);
				int mc = 0, sc = 0;
				room = lt_seed_strand(A, S, Mm, qf, ts, t_len, tab, tsh, k, tot, mc, sc);
				base = 0; n = mc; H.rc = qf.rc;
			} else {
				// anker_rc: forward strand (skipped when none of its every-k-th k-mers is in the index, preseed), then the reverse
				bool fwd = true;
				if(!A.exhaustive) {
					bool hit = false;
					for(int i0 = lane * k; i0 < q_len; i0 += 64 * k) {
						uint64_t key = 0;
						for(int x = 0; x < k; ++x) key = (x ? (key << 2) : 0ull) | (uint64_t) ((i0 + x < q_len) ? qn(qf, i0 + x) : 0);
						if(key <= 0xFFFFFFFFull && lt_lookup(tab, tsh, (uint32_t) key) != 0) hit = true;
					}
					fwd = __any(hit);
				}
				int score = 0, plen = 0, mc = 0, sc = 0;
				if(fwd) room = lt_seed_strand(A, S, Mm, qf, ts, t_len, tab, tsh, k, tot, mc, sc);
				score = sc; plen = mc;
				QView qr = qf; qr.rc = 1;
				mc = 0; sc = 0;
				if(room) room = lt_seed_strand(A, S, Mm, qr, ts, t_len, tab, tsh, k, tot, mc, sc);
				const int bestScore = max(score, sc);
				if(A.one2one && bestScore < k && bestScore * k < (q_len - k - bestScore)) n = 0;
				else if(bestScore == 0) n = 0;
				else if(bestScore == score) { base = 0; n = plen; H.rc = 0; }
				else { base = plen; n = mc; H.rc = 1; qf.rc = 1; }
			}
			if(!room) { if(lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 3ull); n = 0; }
			go = n > 0 && A.stop != 1;
		}
		wave_sync_hbm();    // the MEMs written by lane 0 are read by all lanes from here on
		unsigned mapQ = 0;
		int start = 0, nc = 0;
		if(go) {
			int best = 0;
			start = lt_chain(A, S, Mm, base, n, q_len, t_len, k, &mapQ, &best);
			if(mapQ < (unsigned) A.mq || best < k || A.stop == 2) go = false;
		}
		if(go) {
			// the chain in order
			const bool lds_next = n <= LT_NEXT_CAP;
			wave_sync();
			if(lane == 0) {
				int c = start;
				for(;;) {
					if(lds_next) S.chain[nc] = (uint16_t) c; else Mm.ch[base + nc] = c;
					if(A.ts && (nc || Mm.qS[base + c])) {
						// trimSeeds (chain.c:493-528; align.c:413): the front of a seed goes back to the problem before it; the first seed
						// keeps it when it starts the read; one base of a seed stays
						const int len = Mm.qE[base + c] - Mm.qS[base + c];
						const int cut = len < A.ts ? len - 1 : A.ts;
						Mm.tS[base + c] += cut; Mm.qS[base + c] += cut;
					}
					++nc;
					const int nx = lds_next ? (int) S.next[c] : Mm.nx[base + c];
					if(!nx || nx <= c || nx >= n || nc >= n) break;        // (links point to later MEMs)
					c = nx;
				}
			}
			nc = __shfl(nc, 0);
			wave_sync_hbm();
			// pass 1: a join too large to score positive fails the read (align.c:468-479)
			bool fail = false;
			for(int l = 1 + lane; l < nc; l += 64) {
				const int ia = base + (lds_next ? (int) S.chain[l - 1] : Mm.ch[base + l - 1]), ib = base + (lds_next ? (int) S.chain[l] : Mm.ch[base + l]);
				fail = fail || lt_join(A, Mm, ia, ib, l, nc, q_len, t_len).fail != 0;
			}
			if(__any(fail)) go = false;
		}
		if(go) {
			const bool lds_next = n <= LT_NEXT_CAP;
			const int np = nc + 1;
			unsigned long long first = 0;
			if(lane == 0) first = atomicAdd(&A.counters[LCI(LC_PROB)], (unsigned long long) np);
			first = __shfl(first, 0);
			if((int64_t) first + np > A.prob_cap) { if(lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 5ull); go = false; }
			else {
				H.first = (int64_t) first; H.n_prob = np; H.mapQ = (int) mapQ; H.status = 1;
				if(lane == 0) atomicAdd(&A.counters[LCI(LC_MEMS)], (unsigned long long) n);
				for(int l0 = 0; l0 < np; l0 += 64) {
					const int l = l0 + lane;
					Join J;
					J.flags = PF_NONE; J.t_l = 0; J.q_l = 0; J.band = 0;
					if(l < np) {
						const int ca = l > 0 ? (lds_next ? (int) S.chain[l - 1] : Mm.ch[base + l - 1]) : 0;
						const int cb = l < nc ? (lds_next ? (int) S.chain[l] : Mm.ch[base + l]) : 0;
						J = lt_join(A, Mm, base + ca, base + cb, l, nc, q_len, t_len);
					}
					const bool dp = l < np && !(J.flags & (PF_NONE | PF_DEGEN_I | PF_DEGEN_D));
					// run slots: one word per column at most
					const int need = dp ? J.t_l + J.q_l + 1 : 0;
					const int incl = wave_scan_incl(need, lane);
					const int total = __shfl(incl, 63);
					unsigned long long rbase = 0;
					if(lane == 63 && total) rbase = atomicAdd(&A.counters[LCI(LC_RUNS)], (unsigned long long) total);
					rbase = __shfl(rbase, 63);
					const bool rfit = (int64_t) rbase + total <= A.runs_cap;
					if(!rfit && lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 6ull);
					if(l < np) {
						LtProb P;
						P.runs = (int64_t) rbase + incl - need; P.read = (int32_t) rr; P.t_s = J.t_s; P.t_l = J.t_l; P.q_s = J.q_s; P.q_l = J.q_l;
						P.k = J.k; P.band = J.band; P.flags = (dp && !rfit) ? PF_NONE : J.flags; P.body = J.body;
						// MEM bases score d[b][b] each (a MEM never holds an N)
						P.body_score = 0;
						if(S.d[0] == S.d[6] && S.d[0] == S.d[12] && S.d[0] == S.d[18]) P.body_score = J.body * S.d[0];
						else for(int x = 0; x < J.body; ++x) { const int b = q2(qf, J.bq + x); P.body_score += S.d[6 * b]; }
						P.score = 0; P.n_runs = 0; P.clip = 0; P.pad = 0;
						if(l == 0) { H.pos0 = Mm.tS[base + (lds_next ? (int) S.chain[0] : Mm.ch[base])] - 1; H.clip0 = J.clip0; }
						if(l == nc) H.qe_trail = J.qe;
						A.prob[first + l] = P;
					}
					{
						// work figures: cells of the DP problems (rows x columns, rows x (band + 1) when banded)
						const long long cells = dp ? (long long) J.t_l * (J.band ? (J.band | 1) + 1 : J.q_l) : 0;
						long long tot = cells;
						for(int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
						if(lane == 0 && tot) atomicAdd(&A.counters[LCI(LC_CELLS)], (unsigned long long) tot);
					}
					// queues per class, one atomic per class and round
					int l_iters = 0;
					const int lcls = dp && rfit ? lt_lane_class(J.q_l, J.t_l, J.band, J.k, A.lane_tq, A.lane_mask, A.lane_turns, &l_iters) : -1;
					{
						// the lane classes share one queue (sorted by class and sweep length before their kernels run)
						const unsigned long long m = __ballot(lcls >= 0);
						if(m) {
							const int leader = __ffsll((long long) m) - 1;
							unsigned long long qb = 0;
							if(lane == leader) qb = atomicAdd(&A.counters[LCI(LC_LANE)], (unsigned long long) __popcll(m));
							qb = __shfl(qb, leader);
							if(lcls >= 0) {
								const size_t at = (size_t) qb + __popcll(m & ((1ull << lane) - 1ull));
								A.lq[at] = (int32_t) (first + l);
								A.lkey[at] = ((uint32_t) lcls << 24) | (uint32_t) min(l_iters, 0xFFFFFF);
							}
							for(int c = 0; c < LT_LCLS; ++c) {
								const unsigned long long mc = __ballot(lcls == c);
								if(mc && lane == __ffsll((long long) mc) - 1) atomicAdd(&A.counters[LCI(LC_LCNT + c)], (unsigned long long) __popcll(mc));
							}
						}
					}
					const int cls = dp && rfit && lcls < 0 ? lt_class(J.q_l, J.t_l, J.band, J.k, A.xe_cap) : -1;
					for(int c = 0; c < LT_NCLS; ++c) {
						const unsigned long long m = __ballot(cls == c);
						if(!m) continue;
						const int leader = __ffsll((long long) m) - 1;
						unsigned long long qb = 0;
						if(lane == leader) qb = atomicAdd(&A.counters[LCI(LC_CNT + c)], (unsigned long long) __popcll(m));
						qb = __shfl(qb, leader);
						if(cls == c) A.queue[(size_t) c * A.prob_cap + qb + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t) (first + l);
					}
				}
				// pos0 / clip0 live in lane 0 (l == 0), qe_trail in the lane that owned l == nc
				const int own = nc & 63;
				H.qe_trail = __shfl(H.qe_trail, own);
				H.pos0 = __shfl(H.pos0, 0); H.clip0 = __shfl(H.clip0, 0);
			}
		}
		if(!go) { H.status = 0; H.n_prob = 0; }
		if(lane == 0) A.rd[rr] = H;
		wave_sync();
	}
}

// ---- the alignment columns of one problem as runs ((length << 2) | class: 0 '=', 1 'X', 2 'I' gap in the template, 3 'D' gap
// in the read), written by the lane that walks the move matrix ---------------------------------------------------------------
struct RunOut {
	uint32_t *dst;
	int cap;               // slot words
	int n;                 // runs written
	int cur_cls, cur_len;  // open run
	int total;             // columns
	int first_cls;
	int n_diag;            // runs up to and including the last aligned pair (the open run counted)
	int suf_I;             // 'I' columns behind the last aligned pair
	bool writer;           // false: this lane only follows (a walk that every lane of the wavefront runs in step, lane 0 writing)
	__device__ __forceinline__ void init(uint32_t *d, int c, bool w = true) { dst = d; cap = c; n = 0; cur_cls = -1; cur_len = 0; total = 0; first_cls = -1; n_diag = 0; suf_I = 0; writer = w; }
	__device__ __forceinline__ void put(int cls, int len) {
		if(len <= 0) return;
		if(total == 0) first_cls = cls;
		total += len;
		if(cls == cur_cls) cur_len += len;
		else {
			if(cur_len && n < cap) { if(writer) dst[n] = ((uint32_t) cur_len << 2) | (uint32_t) cur_cls; ++n; }
			cur_cls = cls; cur_len = len;
		}
		if(cls < 2) { n_diag = n + 1; suf_I = 0; }
		else if(cls == 2) suf_I += len;
	}
	// gap columns at the very end of the template are trimmed, the first column always stays (trailTailAln, align.c:180-196);
	// returns the number of trimmed 'I' columns (query bases that become a soft clip)
	__device__ __forceinline__ int finish(bool trail_trim) {
		if(cur_len && n < cap) { if(writer) dst[n] = ((uint32_t) cur_len << 2) | (uint32_t) cur_cls; ++n; }
		cur_len = 0;
		if(!trail_trim || n == n_diag) return 0;
		if(n_diag > 0) { n = n_diag; return suf_I; }
		// nothing but gaps: one column of the first run stays
		if(writer) dst[0] = (1u << 2) | (uint32_t) first_cls;
		n = 1;
		return suf_I - (first_cls == 2 ? 1 : 0);
	}
};

// the walk of the move matrix (nw.c:256-305 full, :586-635 band) from cell (m, n). pitch = bytes per template row, dn = column
// change per template step (0 full matrix, -1 band). A gap run ends on the first cell with EITHER may-open bit (0x30); bit 6 of
// a diagonal cell = the pair is a mismatch. lead: gaps in front are dropped (leadTailAln with t_s == 0, align.c:97-112), a
// dropped 'I' column leaves a read base unaligned (-> clip). Returns the query columns consumed (from q_pos on).
// Compiled as a function of its own (not inlined): inlined into the kernels' divergent regions (one lane per problem walks while
// the others wait) the loop came out of the compiler as one that never ends on gfx950; as a separate function its control flow
// is a single loop. E is a generic pointer (LDS or HBM).
// il != 0: the matrix lies in 16-byte chunks interleaved over the 64 lanes of its wavefront (lt_reg_kernel / lt_regband_kernel: byte b of
// the lane's matrix at ((b >> 4) * 64 + lane) * 16 + (b & 15), E pointing at the lane's first chunk); rows then run downwards (pitch < 0).
__attribute__((noinline)) __device__ int lt_walk(const uint8_t *E, int pitch, int m, int n, int dn, int q_pos, int lead, RunOut *Rp, int *clip, int64_t limit, int *bad, int il = 0) {
	// one flat loop, one cell per iteration: mode 0 = at a cell whose move decides, 1 = inside a run of gaps in the read (template
	// rows consumed), 2 = inside a run of gaps in the template (query columns consumed); a run counts its cells up to and including
	// the first one that carries a may-open bit. A correct matrix is left through one of its zero cells; anything else (leaving the
	// matrix, more iterations than cells) is a bug and must not hang the wavefront.
	RunOut R = *Rp;
	int64_t at = (int64_t) m * pitch + n;
	int mode = 0, g = 0, cl = *clip, isbad = 0;
	bool stop = false;
	// Nine steps in ten are diagonal, and each was a load that waited for the one before it to say where to go (a quarter of a microsecond
	// through L2, a thousand times a read): the next three cells down the diagonal are asked for ahead, so that a diagonal run has four
	// loads in flight; a gap step leaves the diagonal and asks again.
	const int64_t ds = (int64_t) pitch + 1 + dn;
	auto cell = [&](int64_t i) -> int { return (i >= 0 && i < limit) ? (int) E[il ? (((i >> 4) << 10) | (i & 15)) : i] : 0; };
	int e = cell(at), d1 = cell(at + ds), d2 = cell(at + 2 * ds), d3 = cell(at + 3 * ds);
	for(int64_t it = 0; !stop; ++it) {
		if(at < 0 || at >= limit || it > limit) { isbad = 1; stop = true; }
		else {
			bool diag = false;
			if(mode == 0 && e == 0) stop = true;
			else if(mode == 0 && (e & 7) == 1) {
				R.put((e & 64) ? 1 : 0, 1);
				lead = 0;
				at += ds; ++q_pos;
				diag = true;
			} else {
				if(mode == 0) { mode = ((e & 7) >= 4) ? 1 : 2; g = 0; }
				++g;
				const bool last = (e & 0x30) != 0;
				if(mode == 1) {
					at += pitch + dn;
					if(last) { if(!lead) R.put(3, g); mode = 0; }
				} else {
					at += 1;
					if(last) { q_pos += g; if(!lead) R.put(2, g); else cl += g; mode = 0; }
				}
			}
			if(!stop) {
				if(diag) { e = d1; d1 = d2; d2 = d3; d3 = cell(at + 3 * ds); }
				else { e = cell(at); d1 = cell(at + ds); d2 = cell(at + 2 * ds); d3 = cell(at + 3 * ds); }
			}
		}
	}
	*Rp = R; *clip = cl; *bad = isbad;
	return q_pos;
}

struct DpLds {
	int d[25];
	uint8_t E[4][LT_E_WAVE];
	uint8_t t[4][8 * (LT_TMAX + 1)];
};

// NW / NW with free ends (nw.c:26-309; fill identical to NW_score, nw.c:642-890) for the problems of one size class: W lanes
// per problem, lane n owns query column n, the segment sweeps the anti-diagonals (cell (m, n + 1) from the right neighbour's
// last step, (m + 1, n + 1) from the step before), every cell's move byte goes to the problem's matrix in LDS, then lane 0 of
// the segment walks it.
template <int W>
__global__ __launch_bounds__(256) void lt_dp_kernel(const LtArgs A, int cls) {
	constexpr int G = 64 / W;
	__shared__ DpLds S;
	if(threadIdx.x < 25) S.d[threadIdx.x] = A.d[threadIdx.x];
	__syncthreads();
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n = lane & (W - 1), g = lane / W;
	uint8_t *const E = S.E[wave] + g * (LT_E_WAVE / G);
	uint8_t *const tbuf = S.t[wave] + g * (LT_TMAX + 1);
	const unsigned long long count = A.counters[LCI(LC_CNT + cls)];
	const int32_t *queue = A.queue + (size_t) cls * A.prob_cap;
	const int U = A.U, W1 = A.W1;
	for(unsigned long long base = ((unsigned long long) blockIdx.x * 4 + wave) * G; base < count; base += (unsigned long long) gridDim.x * 4 * G) {
		const bool live = base + g < count;
		LtProb *P = A.prob + queue[live ? base + g : base];
		const int k = P->k, t_s = P->t_s, t_len = P->t_l, q_s = P->q_s, q_len = P->q_l, flags = P->flags;
		const int64_t r = A.r0 + P->read;
		const int tt = A.tmpl ? A.tmpl[r] : A.tmpl_all;
		const int at = abs(tt);
		const int tlen_total = A.db.tlen[at];
		const uint64_t *ts = A.db.tseq + A.db.tseq_off[at];
		QView q;
		q.w = A.seq + A.seq_off[r]; q.L = A.len[r]; q.rc = A.rd[P->read].rc;
		q.N = A.N + A.N_off[r]; q.nN = (int) (A.N_off[r + 1] - A.N_off[r]);
		const int pitch = q_len + 1;
		const int low = (t_len + q_len) * (A.MM + U + W1);
		wave_sync();           // the walk of the round before is done with E
		if(live) {
			for(int i = n; i < t_len; i += W) {
				int pos = t_s + i;
				if(pos >= tlen_total) pos -= tlen_total;
				tbuf[i] = (uint8_t) tn(ts, pos);
			}
			// boundary cells (nw.c:703-750): column q_len of every row, row t_len
			for(int m = n; m < t_len; m += W) E[m * pitch + q_len] = (0 < k) ? 0 : ((m == t_len - 1) ? 36 : 5);
			uint8_t *Er = E + t_len * pitch;
			for(int c = n; c <= q_len; c += W) Er[c] = (k == 2 || c == q_len) ? 0 : ((c == q_len - 1) ? 18 : 3);
		}
		const bool col = live && n < q_len;
		const int qc = col ? qn(q, q_s + n) : 0;
		wave_sync();
		// `l*` = the row this lane computed last (start: boundary row m = t_len), bD = D of the row before it
		int lD, lP = low, lQ = low, bD = 0;
		if(k == 2) lD = 0; else lD = W1 + (q_len - 1 - n) * U;
		int best = low, best_m = 0;
		int steps = live ? t_len + q_len - 1 : 0;
		steps = wave_max(steps);
		for(int dstep = 0; dstep < steps; ++dstep) {
			int rD = __shfl_down(lD, 1, W), rQ = __shfl_down(lQ, 1, W), dD = __shfl_down(bD, 1, W);
			const int i = dstep - (q_len - 1 - n);
			if(col && i >= 0 && i < t_len) {
				const int m = t_len - 1 - i;
				if(n == q_len - 1) {
					// boundary column q_len
					rD = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
					rQ = low;
					if(m + 1 == t_len) dD = 0;
					else dD = (0 < k) ? 0 : (W1 + (t_len - 2 - m) * U);
				}
				const int tb = (int) tbuf[m];
				const int *drow = S.d + 5 * tb;
				int Q = rD + W1, Pn = lD + W1, D, mv, cell = 0;
				if(Q < Pn) { D = Pn; mv = 4; } else { D = Q; mv = 2; }
				int x = rQ + U;
				if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
				x = lP + U;
				if(Pn < x) { Pn = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
				x = dD + drow[qc];
				if(D <= x) { D = x; cell |= 1 | (tb != qc ? 64 : 0); } else cell |= mv;
				E[m * pitch + n] = (uint8_t) cell;
				bD = lD;
				lD = D; lP = Pn; lQ = Q;
				if(n == 0 && k < 0 && best < D) { best = D; best_m = m; }
			}
		}
		wave_sync();
		// result selection (nw.c:218-254): k < 0: best cell of column 0 (rows from the bottom up, strict); k == -2: then the
		// cells of row 0, `<=`, so the largest column holding the row maximum if it reaches the score
		int score, sm = 0, sn = 0;
		const int seg0 = g * W;
		if(k < 0) { score = __shfl(best, seg0); sm = __shfl(best_m, seg0); }
		else score = __shfl(lD, seg0);
		{
			int mx = col ? lD : INT_MIN;
			for(int o = W / 2; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, W));
			const unsigned long long who = __ballot(col && lD == mx);
			const unsigned long long mine = (W == 64) ? who : ((who >> seg0) & ((1ull << (W & 63)) - 1ull));
			if(k == -2 && mx >= score && mine) { score = mx; sm = 0; sn = 63 - __clzll((long long) mine); }
		}
		if(live && n == 0) {
			RunOut R;
			R.init(A.runs + P->runs, t_len + q_len + 1);
			int clip = sn, bad = 0;
			const int q_pos = lt_walk((const uint8_t *) E, pitch, sm, sn, 0, sn, (flags & PF_LEAD_TRIM) != 0, &R, &clip, (int64_t) pitch * (t_len + 1), &bad);
			const int cut = R.finish((flags & PF_TRAIL_TRIM) != 0);
			if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
			P->score = score; P->n_runs = bad ? 0 : R.n;
			P->clip = (k > 0) ? (q_len - q_pos + cut) : clip;
		}
	}
}

// ---- problems of more than 64 query columns and banded problems --------------------------------------------------------------
// One problem per wavefront, lane n owns XW neighbouring columns (aligned to the right end, so only lane 0 can own fewer) and
// takes them right to left inside a step of the same anti-diagonal sweep. banded: NW_band (nw.c:310-640). In query coordinates
// the banded recurrences are the full ones restricted to the columns [c - band/2, c + band/2] of each row (c = (t_len + q_len)/2
// at the last row, one less per row), except that the leftmost cell of a row has no template-gap state (nw.c:505-531), that
// the right neighbour of the band's last column is a virtual cell (D = low, move byte 37) and that the result is read off the
// leftmost cells. The move matrix is stored as the reference stores it: row pitch band + 2, cell (m, column) at index
// column - c_m + band/2, so the walk is the reference's (one column to the left per template row).
template <int XW, bool banded>
__device__ void lt_sweep_x(const LtArgs &A, const int *sd, const LtProb *P, const QView &q, uint8_t *E, const uint8_t *tbuf, const uint64_t *ts,
                           int tlen_total, int &o_score, int &o_m, int &o_n, int &o_qpos) {
	const int lane = threadIdx.x & 63;
	const int k = P->k, t_s = P->t_s, t_len = P->t_l, q_s = P->q_s, q_len = P->q_l;
	int band = banded ? P->band : 0;
	const int U = A.U, W1 = A.W1;
	const int low = (t_len + q_len) * (A.MM + U + W1);
	if(band & 1) ++band;
	const int half = band >> 1, bq = band + 1;
	const int pitch = banded ? bq + 1 : q_len + 1;
	const int cbot = (t_len + q_len) >> 1;             // band centre of the last row (m = t_len - 1)
	// boundary cells: row t_len and, per row, the cell behind the last column
	if(banded) {
		const int sn0 = q_len - (cbot + 1) + half;
		uint8_t *Er = E + (int64_t) t_len * pitch;
		for(int c = lane; c <= sn0; c += 64) Er[c] = (k == 2 || c == sn0) ? 0 : ((c == sn0 - 1) ? 18 : 3);
		for(int m = lane; m < t_len; m += 64) {
			const int c = cbot - (t_len - 1 - m);
			if(c + half < q_len - 1) E[(int64_t) m * pitch + bq] = 37;
			else E[(int64_t) m * pitch + (q_len - c + half)] = (0 < k) ? 0 : 37;
		}
	} else {
		for(int m = lane; m < t_len; m += 64) E[(int64_t) m * pitch + q_len] = (0 < k) ? 0 : ((m == t_len - 1) ? 36 : 5);
		uint8_t *Er = E + (int64_t) t_len * pitch;
		for(int c = lane; c <= q_len; c += 64) Er[c] = (k == 2 || c == q_len) ? 0 : ((c == q_len - 1) ? 18 : 3);
	}
	const int nl = (q_len + XW - 1) / XW;              // lanes in use
	const int c0 = q_len - (nl - lane) * XW;           // this lane's first column (negative: lane 0 owns fewer than XW)
	const bool act = lane < nl;
	int qc[XW], lD[XW], lP[XW], lQ[XW];
#pragma unroll
	for(int j = 0; j < XW; ++j) {
		const int c = c0 + j;
		qc[j] = (act && c >= 0) ? qn(q, q_s + c) : 0;
		if(k == 2) lD[j] = 0; else lD[j] = W1 + (q_len - 1 - c) * U;
		lP[j] = low; lQ[j] = low;
	}
	int bD0 = 0;                                       // the row before `l` of this lane's first column, for the lane to the left
	int best = low, bm = 0;
	wave_sync();
	const int steps = t_len + nl - 1;
	for(int d = 0; d < steps; ++d) {
		const int nD = __shfl_down(lD[0], 1), nQ = __shfl_down(lQ[0], 1), nbD = __shfl_down(bD0, 1);
		const int i = d - (nl - 1 - lane);
		if(act && i >= 0 && i < t_len) {
			const int m = t_len - 1 - i;
			const int c = cbot - i;
			const int eq = banded ? max(c - half, 0) : -1;
			const bool virt = banded && c + half < q_len - 1;
			const int sq = virt ? c + half : q_len - 1;
			int rD, rQ, dD;
			const int cr = c0 + XW;                         // first column of the lane to the right (q_len for the last lane)
			if(cr >= q_len) {
				rD = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
				rQ = low;
				if(m + 1 == t_len) dD = 0;
				else dD = (0 < k) ? 0 : (W1 + (t_len - 2 - m) * U);
			} else if(cr <= sq) { rD = nD; rQ = nQ; dD = nbD; }      // it has done this row already
			else { rD = low; rQ = low; dD = nD; }                      // outside the band: virtual cell, `l` is the row below
			int tb;
			if(m < LT_XT_LDS) tb = (int) tbuf[m];
			else { int pos = t_s + m; if(pos >= tlen_total) pos -= tlen_total; tb = tn(ts, pos); }
			const int *drow = sd + 5 * tb;
			uint8_t *e = E + (int64_t) m * pitch + (banded ? half - c : 0);
#pragma unroll
			for(int jj = 0; jj < XW; ++jj) {
				const int j = XW - 1 - jj;
				const int col = c0 + j;
				if(col >= 0 && col >= eq && col <= sq) {
					const int oD = lD[j];
					int D, Q, Pn, cell = 0;
					if(col == eq) {
						// leftmost cell of the band: no template-gap state
						int mv;
						Q = rD + W1;
						const int x = rQ + U;
						if(Q < x) { Q = x; mv = 3; } else { mv = 2; cell |= 16; }
						D = dD + drow[qc[j]];
						if(Q <= D) cell |= 1 | (tb != qc[j] ? 64 : 0); else { D = Q; cell |= mv; }
						Pn = low;
						if(eq == 0 && k < 0 && best < D) { best = D; bm = m; }
					} else {
						int mv;
						Q = rD + W1; Pn = oD + W1;
						if(Q < Pn) { D = Pn; mv = 4; } else { D = Q; mv = 2; }
						int x = rQ + U;
						if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
						x = lP[j] + U;
						if(Pn < x) { Pn = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
						x = dD + drow[qc[j]];
						if(D <= x) { D = x; cell |= 1 | (tb != qc[j] ? 64 : 0); } else cell |= mv;
						if(!banded && col == 0 && k < 0 && best < D) { best = D; bm = m; }
					}
					e[col] = (uint8_t) cell;
					if(j == 0) bD0 = oD;
					lD[j] = D; lP[j] = Pn; lQ[j] = Q;
					rD = D; rQ = Q; dD = oD;
				} else if(col >= 0 && col == sq + 1) {
					rD = low; rQ = low; dD = lD[j];
				}
			}
		}
	}
	wave_sync();
	// result selection. Full: nw.c:218-254, banded: nw.c:557-585. Column `cres` (0, or the leftmost column of the last row's
	// band) holds the final value; lane 0 tracked the best leftmost cell for k < 0.
	const int cfin = cbot - (t_len - 1);
	const int cres = banded ? max(cfin - half, 0) : 0;
	const int sfin = (banded && cfin + half < q_len - 1) ? cfin + half : q_len - 1;
	const int owner = nl - 1 - (q_len - 1 - cres) / XW;
	int Dres = 0;
#pragma unroll
	for(int j = 0; j < XW; ++j) if(c0 + j == cres) Dres = lD[j];
	Dres = __shfl(Dres, owner);
	int score = __shfl(best, 0), row = __shfl(bm, 0), colr = 0;
	if(banded) { if(row == 0) { score = Dres; colr = cres; } }
	else if(!(k < 0)) { score = Dres; row = 0; }
	int q_pos = 0;
	if(k == -2) {
		int mx = INT_MIN, mc = -1;
#pragma unroll
		for(int j = 0; j < XW; ++j) if(act && c0 + j >= cres && c0 + j <= sfin && lD[j] >= mx) { mx = lD[j]; mc = c0 + j; }
		const int wmx = wave_max(mx);
		const unsigned long long who = __ballot(act && mc >= 0 && mx == wmx);
		const int src = who ? 63 - __clzll((long long) who) : 0;
		const int mcs = __shfl(mc, src);
		if(wmx >= score) { score = wmx; row = 0; colr = mcs; q_pos = mcs - cres; }
	}
	o_score = score; o_m = row;
	o_n = banded ? colr - (cbot - (t_len - 1 - row)) + half : colr;
	o_qpos = q_pos;
}

// One lane, rows and move matrix in HBM: NW / NW_band as the reference runs them (nw.c:26-640), for what fits neither sweep
// (more than 255 query columns, a banded problem whose last-row scan would read cells the band no longer covers).
__device__ bool lt_serial(const LtArgs &A, const int *sd, const LtProb *P, const QView &q, const uint64_t *ts, int tlen_total,
                          uint8_t *E, int64_t e_cap, int32_t *rows, int ncols, int &o_score, int &o_m, int &o_n, int &o_qpos, int &o_pitch, int &o_dn) {
	const int k = P->k, t_s = P->t_s, t_len = P->t_l, q_s = P->q_s, q_len = P->q_l;
	int band = P->band;
	const int U = A.U, W1 = A.W1;
	const int low = (t_len + q_len) * (A.MM + U + W1);
	int32_t *Dr[2] = {rows, rows + ncols}, *Pr[2] = {rows + 2 * (int64_t) ncols, rows + 3 * (int64_t) ncols};
	int dc = 0, dp = 1;
	const int t_e = t_s + t_len;      // may exceed the template: positions are taken modulo its length
	int score = low;
	if(band == 0) {
		const int pitch = q_len + 1;
		if((int64_t) pitch * (t_len + 1) > e_cap || q_len + 2 > ncols) return false;
		uint8_t *Er = E + (int64_t) pitch * t_len;
		for(int m = 0; m < t_len; ++m) E[(int64_t) pitch * m + q_len] = (0 < k) ? 0 : 5;
		if(!(0 < k)) E[(int64_t) pitch * (t_len - 1) + q_len] = 36;
		if(k == 2) { for(int n = q_len; n >= 0; --n) { Dr[dp][n] = 0; Pr[dp][n] = low; Er[n] = 0; } }
		else {
			for(int n = q_len - 1; n >= 0; --n) { Dr[dp][n] = W1 + (q_len - 1 - n) * U; Pr[dp][n] = low; Er[n] = 3; }
			Er[q_len - 1] = 18; Er[q_len] = 0; Dr[dp][q_len] = 0; Pr[dp][q_len] = 0;
		}
		int best_m = 0, npos = t_e - 1;
		if(npos >= tlen_total) npos -= tlen_total;
		for(int m = t_len - 1; m >= 0; --m, --npos) {
			if(npos < 0) npos = tlen_total - 1;
			uint8_t *e = E + (int64_t) pitch * m;
			Dr[dc][q_len] = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
			int Qprev = low;
			const int tb = tn(ts, npos);
			for(int n = q_len - 1; n >= 0; --n) {
				int cell = 0, mv;
				int Q = Dr[dc][n + 1] + W1, Pn = Dr[dp][n] + W1, D;
				if(Q < Pn) { D = Pn; mv = 4; } else { D = Q; mv = 2; }
				int x = Qprev + U;
				if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
				x = Pr[dp][n] + U;
				if(Pn < x) { Pn = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
				const int qb = qn(q, q_s + n);
				x = Dr[dp][n + 1] + sd[5 * tb + qb];
				if(D <= x) { D = x; cell |= 1 | (tb != qb ? 64 : 0); } else cell |= mv;
				Dr[dc][n] = D; Pr[dc][n] = Pn; e[n] = (uint8_t) cell; Qprev = Q;
			}
			if(k < 0 && score < Dr[dc][0]) { score = Dr[dc][0]; best_m = m; }
			dc ^= 1; dp ^= 1;
		}
		int sm = 0, sn = 0;
		if(k < 0) {
			sm = best_m;
			if(k == -2) for(int n = 0; n < q_len; ++n) if(score <= Dr[dp][n]) { score = Dr[dp][n]; sm = 0; sn = n; }
		} else score = Dr[dp][0];
		o_score = score; o_m = sm; o_n = sn; o_qpos = sn; o_pitch = pitch; o_dn = 0;
		return true;
	}
	if(band & 1) ++band;
	const int half = band >> 1, bq = band + 1, pitch = bq + 1;
	if((int64_t) pitch * (t_len + 1) > e_cap || band + 4 > ncols) return false;
	uint8_t *Er = E + (int64_t) pitch * t_len;
	int c = (t_len + q_len) >> 1;
	int sn = q_len - 1 - (c - half);
	for(int n = 0; n <= bq; ++n) { Dr[0][n] = 0; Dr[1][n] = 0; Pr[0][n] = 0; Pr[1][n] = 0; }
	if(k != 2) {
		for(int n = sn - 1; n >= 0; --n) { Dr[dp][n] = W1 + (sn - n - 1) * U; Pr[dp][n] = low; Er[n] = 3; }
		Er[sn - 1] = 18; Er[sn] = 0; Dr[dp][sn] = 0; Pr[dp][sn] = 0;
	} else {
		for(int n = sn; n >= 0; --n) { Dr[dp][n] = 0; Pr[dp][n] = low; Er[n] = 0; }
	}
	int bm = 0, bn = 0, en = 0, n = 0, npos = t_e - 1;
	if(npos >= tlen_total) npos -= tlen_total;
	for(int m = t_len - 1; m >= 0; --m, --npos, --c) {
		if(npos < 0) npos = tlen_total - 1;
		uint8_t *e = E + (int64_t) pitch * m;
		int sq = c + half, eq = c - half;
		if(eq < 0) { eq = 0; ++en; } else en = 0;
		int Qprev = low;
		if(sq < q_len - 1) { sn = bq - 1; Dr[dc][bq] = low; e[bq] = 37; }
		else {
			sq = q_len - 1; sn = en + (q_len - eq);
			Dr[dc][sn] = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
			e[sn] = (0 < k) ? 0 : 37;
			--sn;
		}
		const int tb = tn(ts, npos);
		int qp = sq;
		for(n = sn; n > en; --qp, --n) {
			int cell = 0, mv;
			int Q = Dr[dc][n + 1] + W1, Pn = Dr[dp][n - 1] + W1, D;
			if(Q < Pn) { D = Pn; mv = 4; } else { D = Q; mv = 2; }
			int x = Qprev + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
			x = Pr[dp][n - 1] + U;
			if(Pn < x) { Pn = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
			const int qb = qn(q, q_s + qp);
			x = Dr[dp][n] + sd[5 * tb + qb];
			if(D <= x) { D = x; cell |= 1 | (tb != qb ? 64 : 0); } else cell |= mv;
			Dr[dc][n] = D; Pr[dc][n] = Pn; e[n] = (uint8_t) cell; Qprev = Q;
		}
		{	// band edge: no template-gap state
			int cell = 0, mv;
			int Q = Dr[dc][n + 1] + W1, x = Qprev + U;
			if(Q < x) { Q = x; mv = 3; } else { mv = 2; cell |= 16; }
			Pr[dc][n] = low;
			const int qb = qn(q, q_s + qp);
			int D = Dr[dp][n] + sd[5 * tb + qb];
			if(Q <= D) cell |= 1 | (tb != qb ? 64 : 0); else { D = Q; cell |= mv; }
			Dr[dc][n] = D; e[n] = (uint8_t) cell;
		}
		if(eq == 0 && k < 0 && score < Dr[dc][n]) { score = Dr[dc][n]; bm = m; bn = n; }
		dc ^= 1; dp ^= 1;
	}
	int q_pos = 0;
	if(bm == 0) { bn = en; score = Dr[dp][en]; }
	if(k == -2) for(n = en; n < bq; ++n) if(score <= Dr[dp][n]) { score = Dr[dp][n]; bm = 0; bn = n; q_pos = n - en; }
	o_score = score; o_m = bm; o_n = bn; o_qpos = q_pos; o_pitch = pitch; o_dn = -1;
	return true;
}

struct DpxLds {
	int d[25];
	uint8_t t[LT_XT_LDS];
};

#define LT_U(x) __builtin_amdgcn_readfirstlane(x)
// the same value, but opaque to the compiler's uniformity analysis: what depends on it is kept in vector registers. These
// kernels hold one problem per wavefront, so every problem field is wave-uniform and would be kept in scalar registers -- more of
// them than there are, and the builds that spilled scalar registers (into lanes of a vector register) hung on gfx950.
__device__ __forceinline__ int lt_vgpr(int x) { asm volatile("" : "+v"(x)); return x; }

// one problem of class (EHBM ? 9 : 4) + (banded ? 2 : 0) + (XW == 4) per wavefront (XW 8 / 16: 13 + (banded ? 2 : 0) + (XW == 16),
// always EHBM); EHBM: the move matrix in the workgroup's HBM scratch instead of LDS (a kernel of its own, not a branch: see the note
// at lt_walk)
template <int XW, bool banded, bool EHBM>
__global__ __launch_bounds__(64) void lt_dpx_kernel(const LtArgs A) {
	// (a class of its own with the matrix of the 256..511-column problems in 96 KB of LDS was tried: its 2.5 ms instead of the 12.8 ms
	// of the HBM form cost the whole stage 10 ms -- a workgroup that takes most of a CU's LDS keeps the lane kernels off that CU, the
	// HBM form runs beside them for nothing)
	static_assert(XW <= 4 || EHBM, "the wide classes keep their move matrix in HBM");
	constexpr int cls = XW <= 4 ? (EHBM ? 9 : 4) + (banded ? 2 : 0) + (XW == 4 ? 1 : 0) : 13 + (banded ? 2 : 0) + (XW == 16 ? 1 : 0);
	__shared__ DpxLds S;
	extern __shared__ uint32_t lt_dpx_e[];      // the move matrix of the LDS forms: LT_XE_LDS bytes
	uint8_t *const Ebuf = EHBM ? A.xE + (size_t) blockIdx.x * ((size_t) A.xe_cap + (size_t) 16 * A.xrow) : (uint8_t *) lt_dpx_e;
	const int lane = threadIdx.x;
	if(lane < 25) S.d[lane] = A.d[lane];
	wave_sync();
	const unsigned long long count = A.counters[LCI(LC_CNT + cls)];
	const int32_t *queue = A.queue + (size_t) cls * A.prob_cap;
	for(unsigned long long idx = blockIdx.x; idx < count; idx += gridDim.x) {
		LtProb *P = A.prob + lt_vgpr(queue[idx]);
		const int k = P->k, t_s = P->t_s, t_len = P->t_l, q_len = P->q_l, flags = P->flags;
		int band = P->band;
		if(lane == 0) { LT_REC(0, idx); LT_REC(1, 1); LT_REC(2, k); LT_REC(3, t_len); LT_REC(4, q_len); LT_REC(5, band); }
		const int64_t r = A.r0 + P->read;
		const int tt = A.tmpl ? A.tmpl[r] : lt_vgpr(A.tmpl_all);
		const int at = abs(tt);
		const int tlen_total = A.db.tlen[at];
		const uint64_t *ts = A.db.tseq + A.db.tseq_off[at];
		QView q;
		q.w = A.seq + A.seq_off[r]; q.L = A.len[r]; q.rc = A.rd[P->read].rc;
		q.N = A.N + A.N_off[r]; q.nN = (int) (A.N_off[r + 1] - A.N_off[r]);
		if(band & 1) ++band;
		const int pitch = banded ? band + 2 : q_len + 1;
		const int64_t need = (int64_t) pitch * (t_len + 1);
		int score = 0, sm = 0, sn = 0, q_pos = 0;
		wave_sync();
		for(int i = lane; i < t_len && i < LT_XT_LDS; i += 64) {
			int pos = t_s + i;
			if(pos >= tlen_total) pos -= tlen_total;
			S.t[i] = (uint8_t) tn(ts, pos);
		}
		wave_sync();
		lt_sweep_x<XW, banded>(A, S.d, P, q, Ebuf, S.t, ts, tlen_total, score, sm, sn, q_pos);
		if(EHBM) wave_sync_hbm();
		const uint8_t *E = (const uint8_t *) Ebuf;
		if(lane == 0) {
			LT_REC(1, 4);
			RunOut R;
			R.init(A.runs + P->runs, t_len + q_len + 1);
			int clip = q_pos, bad = 0;
			const int qend = lt_walk(E, pitch, sm, sn, banded ? -1 : 0, q_pos, (flags & PF_LEAD_TRIM) != 0, &R, &clip, need, &bad);
			const int cut = R.finish((flags & PF_TRAIL_TRIM) != 0);
			if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
			P->score = score; P->n_runs = bad ? 0 : R.n;
			P->clip = (k > 0) ? (q_len - qend + cut) : clip;
			LT_REC(1, 5);
		}
	}
	if(lane == 0) LT_REC(1, 99);
}

// class 8: what fits neither sweep, the reference's own formulation with rows and move matrix in the workgroup's HBM scratch. All
// 64 lanes run the same instructions on the same addresses (uniform control flow; identical stores coalesce into one): it is one
// lane's worth of work either way, and rare.
__global__ __launch_bounds__(64) void lt_serial_kernel(const LtArgs A) {
	__shared__ int s_d[25];
	const int lane = threadIdx.x;
	if(lane < 25) s_d[lane] = A.d[lane];
	wave_sync();
	const unsigned long long count = A.counters[LCI(LC_CNT + 8)];
	const int32_t *queue = A.queue + (size_t) 8 * A.prob_cap;
	uint8_t *const xE = A.xE + (size_t) blockIdx.x * ((size_t) A.xe_cap + (size_t) 16 * A.xrow);
	int32_t *const xrows = (int32_t *) (xE + A.xe_cap);
	for(unsigned long long idx = blockIdx.x; idx < count; idx += gridDim.x) {
		LtProb *P = A.prob + LT_U(queue[idx]);
		LtProb Pu;      // the problem, wave-uniform
		Pu.runs = P->runs; Pu.read = LT_U(P->read); Pu.t_s = LT_U(P->t_s); Pu.t_l = LT_U(P->t_l); Pu.q_s = LT_U(P->q_s); Pu.q_l = LT_U(P->q_l);
		Pu.k = LT_U(P->k); Pu.band = LT_U(P->band); Pu.flags = LT_U(P->flags);
		const int k = Pu.k, t_len = Pu.t_l, q_len = Pu.q_l, flags = Pu.flags;
		const int64_t r = A.r0 + Pu.read;
		const int tt = A.tmpl ? LT_U(A.tmpl[r]) : A.tmpl_all;
		const int at = abs(tt);
		const int tlen_total = LT_U(A.db.tlen[at]);
		const uint64_t *ts = A.db.tseq + A.db.tseq_off[at];
		QView q;
		q.w = A.seq + A.seq_off[r]; q.L = LT_U(A.len[r]); q.rc = LT_U(A.rd[Pu.read].rc);
		q.N = A.N + A.N_off[r]; q.nN = LT_U((int) (A.N_off[r + 1] - A.N_off[r]));
		int score = 0, sm = 0, sn = 0, q_pos = 0, wpitch = 0, dn = 0;
		const bool ok = lt_serial(A, s_d, &Pu, q, ts, tlen_total, xE, A.xe_cap, xrows, A.xrow, score, sm, sn, q_pos, wpitch, dn);
		wave_sync_hbm();
		if(!LT_U((int) ok)) {
			if(lane == 0) { atomicMax(&A.counters[LCI(LC_STATUS)], 8ull); P->score = 0; P->n_runs = 0; P->clip = 0; P->flags |= PF_NONE; }
		} else {
			RunOut R;
			R.init(A.runs + Pu.runs, t_len + q_len + 1, lane == 0);
			int clip = q_pos, bad = 0;
			const int qend = lt_walk((const uint8_t *) xE, wpitch, sm, sn, dn, q_pos, (flags & PF_LEAD_TRIM) != 0, &R, &clip, (int64_t) wpitch * (t_len + 1), &bad);
			const int cut = R.finish((flags & PF_TRAIL_TRIM) != 0);
			if(lane == 0) {
				if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
				P->score = score; P->n_runs = bad ? 0 : R.n;
				P->clip = (k > 0) ? (q_len - qend + cut) : clip;
			}
		}
	}
}

// ---- one LANE per problem --------------------------------------------------------------------------------------------------
// The sweeps above give a problem 8 to 64 lanes and keep its move matrix in LDS; what the configuration's reads produce by the
// hundred million are problems of a few hundred to a few thousand cells, for which the fill and drain of an anti-diagonal sweep,
// the walk by one lane and the LDS the matrices take leave the chip mostly idle. Here a lane owns a problem and runs the
// reference's own row-by-row recurrence (nw.c:26-309; banded: nw.c:310-640): the row below as (D, P) pairs of 16 bits in LDS
// (column-major over the lanes: no bank conflicts), the query columns as bytes beside it, the template rows 16 per word, the move
// matrix in the lane's own stretch of HBM scratch, four cells per store, and the walk by the same lane right behind (the matrix is
// read back through L2). The problems of a class arrive sorted by the length of their sweep, so the 64 sweeps of a wavefront end
// together; all lanes run ONE flat loop, a cell per turn, the boundary column being a cell like the others whose values are
// replaced. 16 bits hold every value as long as (rows + columns) x the largest penalty stays below 2^15 (lane_tq).
struct LaneArgs {
	const int32_t *queue;        // the class's problems, longest sweep first
	unsigned long long count;
	int R, RQ, TW;               // cells per row, query columns, template words per lane: the LDS geometry
	uint8_t *E;                  // per workgroup 64 x ecap bytes of move matrix
	int ecap;
	int estride;                 // bytes between the lanes' stretches of E (ecap + room for the padding of lt_lane_kernel)
	int ablate;                  // diagnosis (KMAHIP_LT_ABLATE): 1 no matrix stores, 2 no walk, 4 no sweep
};

__device__ __forceinline__ uint32_t lt_pack16(int D, int Pn) { return ((uint32_t) D & 0xffffu) | ((uint32_t) Pn << 16); }
struct LaneProb { LtProb *P; int k, t_s, t_len, q_s, q_len, flags, band; bool live; };

// a wavefront's next 64 problems: descriptors, the template rows into T (16 per word, first row in the top bits), the query
// columns into QB (one byte each)
template <bool QB4>
__device__ __forceinline__ LaneProb lt_lane_stage(const LtArgs &A, const LaneArgs &L, unsigned long long base, int lane, uint32_t *T, uint8_t *QB) {
	LaneProb X;
	X.live = base + lane < L.count;
	LtProb *P = A.prob + L.queue[X.live ? base + lane : base];
	X.P = P;
	X.k = P->k; X.t_s = P->t_s; X.t_len = P->t_l; X.q_s = P->q_s; X.q_len = P->q_l; X.flags = P->flags; X.band = P->band;
	const bool live = X.live;
	const int t_s = X.t_s, t_len = X.t_len, q_s = X.q_s, q_len = X.q_len;
	const int64_t r = A.r0 + P->read;
	const int tt = A.tmpl ? A.tmpl[r] : A.tmpl_all;
	const int at = abs(tt);
	const int tlen_total = A.db.tlen[at];
	const uint64_t *ts = A.db.tseq + A.db.tseq_off[at];
	QView q;
	q.w = A.seq + A.seq_off[r]; q.L = A.len[r]; q.rc = A.rd[P->read].rc;
	q.N = A.N + A.N_off[r]; q.nN = (int) (A.N_off[r + 1] - A.N_off[r]);
	wave_sync();
	const int tw_n = live ? (t_len + 15) >> 4 : 0;
	const int tw_max = wave_max(tw_n);
	for(int j = 0; j < tw_max; ++j) {
		if(j < tw_n) {
			int pos = t_s + 16 * j;
			if(pos >= tlen_total) pos -= tlen_total;
			uint32_t w = 0;
			if(pos + 16 <= tlen_total) w = (uint32_t) (win2(ts, pos) >> 32);
			else for(int x = 0; x < 16; ++x) { int p2 = pos + x; if(p2 >= tlen_total) p2 -= tlen_total; w |= (uint32_t) tn(ts, p2) << (30 - 2 * x); }
			T[j * 64 + lane] = w;
		}
	}
	const int q_max = wave_max(live ? q_len : 0);
	for(int n0 = 0; n0 < q_max; n0 += 16) {
		if(live && n0 < q_len) {
			// sixteen columns, four bits each
			uint64_t cw = 0;
			if(q.nN == 0 && !q.rc) {
				const uint64_t w = win2(q.w, q_s + n0);
				for(int x = 0; x < 16; ++x) cw |= ((w >> (62 - 2 * x)) & 3ull) << (4 * x);
			} else if(q.nN == 0) {
				// stored positions L - 1 - (q_s + n0 + x), complemented; the window starts at the lowest of them
				int ps = q.L - 1 - (q_s + n0 + 15), sh = 0;
				if(ps < 0) { sh = -ps; ps = 0; }
				const uint64_t w = win2(q.w, ps);
				for(int x = 0; x < 16; ++x) if(15 - x - sh >= 0) cw |= (3ull - ((w >> (62 - 2 * (15 - x - sh))) & 3ull)) << (4 * x);
			} else {
				for(int x = 0; x < 16; ++x) if(n0 + x < q_len) cw |= (uint64_t) qn(q, q_s + n0 + x) << (4 * x);
			}
			if(QB4) { for(int x = 0; x < 16; x += 2) if(n0 + x < q_len) QB[((n0 + x) >> 1) * 64 + lane] = (uint8_t) ((cw >> (4 * x)) & 255ull); }
			else for(int x = 0; x < 16; ++x) if(n0 + x < q_len) QB[(n0 + x) * 64 + lane] = (uint8_t) ((cw >> (4 * x)) & 15ull);
		}
	}
	return X;
}

// SIMPLE: the score of a pair is one of three values (match, mismatch, N in the read), as in every scheme the reference's options
// produce -- then it is computed instead of looked up
template <bool SIMPLE>
__global__ __launch_bounds__(64) void lt_lane_kernel(const LtArgs A, const LaneArgs L) {
	extern __shared__ uint32_t lt_lane_lds[];
	int *const sd = (int *) lt_lane_lds;                         // 32 ints
	uint32_t *const DP = lt_lane_lds + 32;                      // R x 64
	uint32_t *const T = DP + (size_t) L.R * 64;                 // TW x 64
	uint8_t *const QB = (uint8_t *) (T + (size_t) L.TW * 64);   // RQ x 64
	const int lane = threadIdx.x;
	if(lane < 25) sd[lane] = A.d[lane];
	uint8_t *const E = L.E + ((size_t) blockIdx.x * 64 + lane) * (size_t) L.estride;
	const int U = A.U, W1 = A.W1;
	const int dM = A.d[0], dX = A.d[1], dN = A.d[4];
	for(unsigned long long base = (unsigned long long) blockIdx.x * 64; base < L.count; base += (unsigned long long) gridDim.x * 64) {
		const LaneProb X = lt_lane_stage<false>(A, L, base, lane, T, QB);
		LtProb *P = X.P;
		const bool live = X.live;
		const int k = X.k, t_len = X.t_len, q_len = X.q_len, flags = X.flags;
		const int pitch = q_len + 1;
		const int low = (t_len + q_len) * (A.MM + U + W1);
		const int q_max = wave_max(live ? q_len : 0);
		if(live) QB[q_len * 64 + lane] = 0;
		// Move bytes go out in descending address order, four to a store. All lanes emit one byte per turn from the first turn of
		// the boundary row on (a lane with fewer columns than the widest starts with padding above its matrix), and every lane's
		// matrix is shifted by up to three bytes so that its first byte is the top byte of a word: the store is then every fourth
		// turn for all lanes alike, a scalar branch.
		const int pad = q_max - q_len;
		int ea = pitch * (t_len + 1) - 1 + pad;            // next byte, before the shift
		const int delta = (3 - ea) & 3;
		uint8_t *const Em = E + delta;                      // the lane's matrix
		ea += delta;                                        // now in the coordinates of E; (ea & 3) == 3
		uint32_t ew = 0;
		for(int i = 0; i <= q_max; ++i) {
			const int n = q_max - i;
			if(live) {
				int D = 0, code = 0;
				if(n < q_len && k != 2) { D = W1 + (q_len - 1 - n) * U; code = (n == q_len - 1) ? 18 : 3; }
				if(n <= q_len) DP[n * 64 + lane] = lt_pack16(D, low);
				ew |= (uint32_t) code << ((3 - (i & 3)) << 3);
				if((i & 3) == 3) { *(uint32_t *) (E + ea) = ew; ew = 0; }
				--ea;
			}
		}
		// the sweep. Per row: the template base, its five scores as bytes of two registers, the boundary column's value and code
		int m = t_len - 1, n = q_len, right = 0, diag = 0, Qprev = low, score = low, best_m = 0;
		int tb = 0, bnd = (0 < k) ? 0 : W1, bcode = (0 < k) ? 0 : 36;
		uint32_t tw1 = 0;                                   // the template word of the row after this one
		if(live) {
			tb = (int) ((T[((m >> 4) << 6) + lane] >> (30 - ((m & 15) << 1))) & 3u);
			tw1 = T[((max(m - 1, 0) >> 4) << 6) + lane];
		}
		const int iters = live && !(L.ablate & 4) ? pitch * t_len : 0;
		const int it_max = wave_max(iters);
		const int e0 = __builtin_amdgcn_readfirstlane(q_max + 1);
		uint32_t below = live ? DP[n * 64 + lane] : 0u;
		int qb = 0;
		for(int it = 0; it < it_max; ++it) {
			if(it < iters) {
				// what the next turn needs is asked for first: its cell of the row below and its query base
				const bool isz = n == 0;
				const int n1 = isz ? q_len : n - 1;
				const uint32_t below1 = DP[n1 * 64 + lane];
				const int qb1 = (int) QB[n1 * 64 + lane];
				int sc;
				if(SIMPLE) sc = qb == 4 ? dN : (tb == qb ? dM : dX);
				else sc = sd[5 * tb + qb];
				const int Db = (int) (short) (below & 0xffffu), Pb = ((int) below) >> 16;
				int Q = right + W1, Pn = Db + W1;
				int mv = Q < Pn ? 4 : 2;
				int D = max(Q, Pn);
				int x = Qprev + U;
				const bool c1 = Q < x;
				mv = (c1 && D <= x) ? 3 : mv;
				int cell = c1 ? 0 : 16;
				Q = max(Q, x); D = max(D, x);
				x = Pb + U;
				const bool c2 = Pn < x;
				mv = (c2 && D <= x) ? 5 : mv;
				cell |= c2 ? 0 : 32;
				Pn = max(Pn, x); D = max(D, x);
				x = diag + sc;
				cell |= (D <= x) ? (tb != qb ? 65 : 1) : mv;
				D = max(D, x);
				const bool isb = n == q_len;
				D = isb ? bnd : D; Q = isb ? low : Q; cell = isb ? bcode : cell;
				DP[n * 64 + lane] = lt_pack16(D, Pn);
				const int pos = 3 - ((e0 + it) & 3);
				ew |= (uint32_t) cell << (pos << 3);
				if(pos == 0 || it == iters - 1) { if(!(L.ablate & 1)) *(uint32_t *) (E + (ea & ~3)) = ew; ew = 0; }
				--ea;
				if(isz && k < 0 && score < D) { score = D; best_m = m; }
				diag = Db; right = D; Qprev = Q;
				if(isz) {
					--m;
					tb = (int) ((tw1 >> (30 - ((m & 15) << 1))) & 3u);
					tw1 = T[((max(m - 1, 0) >> 4) << 6) + lane];
					bnd = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
					bcode = (0 < k) ? 0 : 5;
				}
				n = n1; below = below1; qb = qb1;
			}
		}
		// result selection (nw.c:218-254)
		int sm = 0, sn = 0;
		if(live) {
			if(k < 0) {
				sm = best_m;
				if(k == -2) for(int n2 = 0; n2 < q_len; ++n2) {
					const int Dn = (int) (short) (DP[n2 * 64 + lane] & 0xffffu);
					if(score <= Dn) { score = Dn; sm = 0; sn = n2; }
				}
			} else score = right;
		}
		wave_sync_hbm();   // the matrix is read back by the lane that wrote it
		if(live && !(L.ablate & 2)) {
			RunOut R;
			R.init(A.runs + P->runs, t_len + q_len + 1);
			int clip = sn, bad = 0;
			const int q_pos = lt_walk((const uint8_t *) Em, pitch, sm, sn, 0, sn, (flags & PF_LEAD_TRIM) != 0, &R, &clip, (int64_t) pitch * (t_len + 1), &bad);
			const int cut = R.finish((flags & PF_TRAIL_TRIM) != 0);
			if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
			P->score = score; P->n_runs = bad ? 0 : R.n;
			P->clip = (k > 0) ? (q_len - q_pos + cut) : clip;
		}
	}
}

// the same for NW_band (nw.c:310-640): a row holds the band's cells, index n = column - (c - band / 2) with c the band's centre, which
// moves one column to the left per row; the cell below index n is index n - 1 of the row before, so a row is still updated in
// place. Per row: the cell to the right of the band (virtual, D = low; or the boundary column where the band reaches it), the
// band's cells, and the leftmost one, which has no template-gap state. Rows are cut at the matrix' first and last column (`en`
// leading indices unused once the band has reached column 0).
template <bool SIMPLE>
__global__ __launch_bounds__(64) void lt_lane_band_kernel(const LtArgs A, const LaneArgs L) {
	extern __shared__ uint32_t lt_lane_lds[];
	int *const sd = (int *) lt_lane_lds;
	uint32_t *const DP = lt_lane_lds + 32;
	uint32_t *const T = DP + (size_t) L.R * 64;
	uint8_t *const QB = (uint8_t *) (T + (size_t) L.TW * 64);
	const int lane = threadIdx.x;
	if(lane < 25) sd[lane] = A.d[lane];
	uint8_t *const E = L.E + ((size_t) blockIdx.x * 64 + lane) * (size_t) L.estride;
	const int U = A.U, W1 = A.W1;
	const int dM = A.d[0], dX = A.d[1], dN = A.d[4];
	constexpr int NEG = -(1 << 28);
	for(unsigned long long base = (unsigned long long) blockIdx.x * 64; base < L.count; base += (unsigned long long) gridDim.x * 64) {
		const LaneProb X = lt_lane_stage<true>(A, L, base, lane, T, QB);
		LtProb *P = X.P;
		const bool live = X.live;
		const int k = X.k, t_len = X.t_len, q_len = X.q_len, flags = X.flags;
		int band = X.band;
		if(band & 1) ++band;
		const int half = band >> 1, bq = band + 1, pitch = bq + 1;
		const int low = (t_len + q_len) * (A.MM + U + W1);
		// the row below the last one (nw.c:386-420): index sn0 is the boundary column
		int c = (t_len + q_len) >> 1;
		const int sn0 = q_len - 1 - (c - half);
		const int r_max = wave_max(live ? bq : 0);
		for(int i = 0; i <= r_max; ++i) if(live && i <= bq) DP[i * 64 + lane] = 0;
		// move bytes go out in descending address order, a word is stored when the next byte belongs to another one
		int ea = -1;
		uint32_t ew = 0;
		auto emit = [&](int addr, int code) {
			if((addr >> 2) != (ea >> 2)) { if(ea >= 0 && !(L.ablate & 1)) *(uint32_t *) (E + (ea & ~3)) = ew; ew = 0; }
			ew |= (uint32_t) code << ((addr & 3) << 3);
			ea = addr;
		};
		for(int i = 0; i <= r_max; ++i) {
			const int n = sn0 - i;
			if(live && n >= 0) {
				int D = 0, code = 0;
				if(n < sn0 && k != 2) { D = W1 + (sn0 - n - 1) * U; code = (n == sn0 - 1) ? 18 : 3; }
				DP[n * 64 + lane] = lt_pack16(D, n == sn0 && k != 2 ? 0 : low);
				emit(pitch * t_len + n, code);
			}
		}
		// row state: en = first index in use, sn = last band cell, the cell at sn + 1 opens the row; qcol0 = column of index 0;
		// clipped: the band reaches the last column, index sn + 1 is the boundary column
		int m = t_len - 1, en = 0, sn = 0, n = 0, qcol0 = 0;
		bool clipped = false;
		auto open_row = [&](int cc, int &en_, int &sn_, int &n_, int &qcol0_, bool &clipped_) {
			int sq = cc + half, eq = cc - half;
			if(eq < 0) { eq = 0; ++en_; } else en_ = 0;
			if(sq < q_len - 1) { sn_ = bq - 1; clipped_ = false; }
			else { sn_ = en_ + (q_len - eq) - 1; clipped_ = true; }
			qcol0_ = cc - half;
			n_ = sn_ + 1;
		};
		open_row(c, en, sn, n, qcol0, clipped);
		int right = 0, diag = 0, Qprev = low, score = low, bm = 0, bn = 0;
		// per row: the template base, what the cell to the right of the band holds
		int tb = 0, fD = 0, fcode = 0;
		uint32_t tw1 = 0;
		if(live) {
			tb = (int) ((T[((m >> 4) << 6) + lane] >> (30 - ((m & 15) << 1))) & 3u);
			tw1 = T[((max(m - 1, 0) >> 4) << 6) + lane];
			if(clipped) { fD = (0 < k) ? 0 : W1; fcode = (0 < k) ? 0 : 37; } else { fD = low; fcode = 37; }
		}
		const int it_max = wave_max(live && !(L.ablate & 4) ? pitch * t_len : 0);
		uint32_t below = live ? DP[max(n - 1, 0) * 64 + lane] : 0u;
		auto qcode = [&](int col) { col = min(max(col, 0), q_len - 1); return (int) ((QB[(col >> 1) * 64 + lane] >> ((col & 1) << 2)) & 15u); };
		int qb = live ? qcode(qcol0 + n) : 0;
		for(int it = 0; it < it_max; ++it) {
			if(live && m >= 0) {
				// the next turn's place first (a new row behind the leftmost cell), then what it needs from LDS
				const bool edge = n == en, first = n == sn + 1;
				int en1 = en, sn1 = sn, n1 = n - 1, qcol01 = qcol0;
				bool clipped1 = clipped;
				if(edge && m > 0) open_row(c - 1, en1, sn1, n1, qcol01, clipped1);    // (behind the first row `en` stays: the result is read off it)
				const uint32_t below1 = DP[max(n1 - 1, 0) * 64 + lane];
				const int qb1 = qcode(qcol01 + n1);
				int sc;
				if(SIMPLE) sc = qb == 4 ? dN : (tb == qb ? dM : dX);
				else sc = sd[5 * tb + qb];
				const int Dbl = (int) (short) (below & 0xffffu);
				const int Db = edge ? NEG : Dbl, Pb = edge ? NEG : ((int) below) >> 16;
				int Q = right + W1, Pn = Db + W1;
				int mv = Q < Pn ? 4 : 2;
				int D = max(Q, Pn);
				int x = Qprev + U;
				const bool c1 = Q < x;
				mv = (c1 && D <= x) ? 3 : mv;
				int cell = c1 ? 0 : 16;
				Q = max(Q, x); D = max(D, x);
				x = Pb + U;
				const bool c2 = Pn < x;
				mv = (c2 && D <= x) ? 5 : mv;
				cell |= (c2 || edge) ? 0 : 32;
				Pn = edge ? low : max(Pn, x); D = max(D, x);
				x = diag + sc;
				cell |= (D <= x) ? (tb != qb ? 65 : 1) : mv;
				D = max(D, x);
				// to the right of the band: a virtual cell, or the boundary column (nw.c:470-500)
				D = first ? fD : D; cell = first ? fcode : cell; Q = first ? low : Q; Pn = first ? low : Pn;
				DP[n * 64 + lane] = lt_pack16(D, Pn);
				emit(pitch * m + n, cell);
				diag = Dbl; right = D; Qprev = Q;
				if(edge) {
					if(qcol0 + en == 0 && k < 0 && score < D) { score = D; bm = m; bn = n; }
					--m; --c;
					tb = (int) ((tw1 >> (30 - ((m & 15) << 1))) & 3u);
					tw1 = T[((max(m - 1, 0) >> 4) << 6) + lane];
					if(clipped1) { fD = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U); fcode = (0 < k) ? 0 : 37; } else { fD = low; fcode = 37; }
				}
				en = en1; sn = sn1; n = n1; qcol0 = qcol01; clipped = clipped1;
				below = below1; qb = qb1;
			}
		}
		if(ea >= 0 && !(L.ablate & 1)) *(uint32_t *) (E + (ea & ~3)) = ew;
		// result selection (nw.c:557-585): the leftmost cell of the first row unless a row above scored higher in column 0
		int q_pos = 0;
		if(live) {
			if(bm == 0) { bn = en; score = right; }
			if(k == -2) for(int n2 = en; n2 < bq; ++n2) {
				const int Dn = (int) (short) (DP[n2 * 64 + lane] & 0xffffu);
				if(score <= Dn) { score = Dn; bm = 0; bn = n2; q_pos = n2 - en; }
			}
		}
		wave_sync_hbm();
		if(live && !(L.ablate & 2)) {
			RunOut R;
			R.init(A.runs + P->runs, t_len + q_len + 1);
			int clip = q_pos, bad = 0;
			const int qend = lt_walk((const uint8_t *) E, pitch, bm, bn, -1, q_pos, (flags & PF_LEAD_TRIM) != 0, &R, &clip, (int64_t) pitch * (t_len + 1), &bad);
			const int cut = R.finish((flags & PF_TRAIL_TRIM) != 0);
			if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
			P->score = score; P->n_runs = bad ? 0 : R.n;
			P->clip = (k > 0) ? (q_len - qend + cut) : clip;
		}
	}
}

// ---- round 4: the DP row in REGISTERS ---------------------------------------------------------------------------------------------
// lt_lane_kernel keeps a lane's row in LDS (an address, a read and a write per cell, 4-16 KB of LDS per wavefront: 1-4 wavefronts per
// SIMD) and takes one flat turn per cell, whose column it has to look up: 72 vector instructions per cell, issued in half of the
// wavefront's cycles (profiles/r3_c4_sq_counters.txt). Here the row is R registers: the loop over a row's columns is unrolled, so the
// column of a cell is a constant -- no LDS traffic for the row, no address arithmetic, the query base by a constant shift out of
// registers filled once per problem, the boundary column and the end of a row straight-line code, four move bytes packed by constant
// shifts per store. A lane's columns are aligned to the RIGHT (column n sits in register n + R - 1 - q_len: the boundary column in the
// last one for every lane), the registers to the left of column 0 compute garbage that nothing reads (a cell depends on the cells
// to its right and below only). The move byte says diagonal / gap in the read / gap in the template, which is all lt_walk asks of
// it ((e & 7) == 1, >= 4, else), with the reference's tie rules: the diagonal wins every tie (nw.c:150-158 `D <= x`), an
// extended gap in the read (P) wins against Q exactly when it beat its own opening (`Pn < x`), an opened one loses to Q.
typedef uint32_t __attribute__((aligned(1))) lt_u32_u;
typedef uint32_t lt_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t lt_u32x4 __attribute__((ext_vector_type(4)));
typedef lt_u32x2 __attribute__((aligned(1))) lt_u32x2_u;
typedef lt_u32x4 __attribute__((aligned(1))) lt_u32x4_u;
// The move bytes of sixteen registers per store (at four bytes a store the sweeps waited for the address path), and the matrices of a
// wavefront's 64 lanes INTERLEAVED in chunks of 16 bytes: a row of a lane's matrix is RP = R rounded up to 16 bytes, indexed by register
// (column n in byte n + off), the rows counted from the boundary row (row 0) in the order they are written -- the same for every lane --
// so that the 64 lanes of a store write one contiguous kilobyte instead of 64 lines of their own. Register j's byte into word
// (j >> 2) & 3; stored when j is the lowest register of its group of sixteen (of eight at the top of a 72-register row). `er` = the
// lane's first chunk of the row.
#define LT_EW_PUT(R_, j_, cell_) do { const uint32_t sh_ = (cell_) << (((j_) & 3) << 3); \
	if((((j_) >> 2) & 3) == 0) ew0 |= sh_; else if((((j_) >> 2) & 3) == 1) ew1 |= sh_; else if((((j_) >> 2) & 3) == 2) ew2 |= sh_; else ew3 |= sh_; \
	if(((j_) & 15) == 0) { \
		if((R_) - (j_) >= 16) { lt_u32x4 w_; w_.x = ew0; w_.y = ew1; w_.z = ew2; w_.w = ew3; *(lt_u32x4 *) (er + (size_t) ((j_) >> 4) * 1024) = w_; } \
		else { lt_u32x2 w_; w_.x = ew0; w_.y = ew1; *(lt_u32x2 *) (er + (size_t) ((j_) >> 4) * 1024) = w_; } \
		ew0 = 0; ew1 = 0; ew2 = 0; ew3 = 0; } } while(0)
__device__ __forceinline__ uint32_t lt_opaque(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
// the new value of a row register INTO that register (the operand is tied): without it the compiler keeps the row before and the row
// being made in registers of their own -- 2 R of them -- and spills; a reload inside the row loop waits for the move matrix' stores
// (same counter) to reach HBM
#define LT_ROW_SET(reg, val) asm volatile("v_mov_b32 %0, %1" : "+v"(reg) : "v"(val))

template <int R>
// (wavefronts per SIMD: what leaves the row loop without a spilled register -- a reload waits for the move matrix' stores, which
// count on the same counter, to reach HBM)
#ifndef LT_REG16_WAVES
#define LT_REG16_WAVES 4
#endif
#ifndef LT_REG32_WAVES
#define LT_REG32_WAVES 4
#endif
#ifndef LT_REG48_WAVES
#define LT_REG48_WAVES 3
#endif
#ifndef LT_REG64_WAVES
#define LT_REG64_WAVES 2
#endif
#ifndef LT_RB72_WAVES
#define LT_RB72_WAVES 2
#endif
__global__ __launch_bounds__(64, (R <= 16 ? LT_REG16_WAVES : R <= 32 ? LT_REG32_WAVES : R <= 48 ? LT_REG48_WAVES : R <= 64 ? LT_REG64_WAVES : 1)) void lt_reg_kernel(const LtArgs A, const LaneArgs L) {
	extern __shared__ uint32_t lt_lane_lds[];
	uint32_t *const T = lt_lane_lds + 32;                        // TW x 64 template words
	uint8_t *const QB = (uint8_t *) (T + (size_t) L.TW * 64);   // RQ x 64 query codes
	const int lane = threadIdx.x;
	constexpr int RP = (R + 15) & ~15, CH = RP / 16;             // bytes and chunks per row of the interleaved matrix
	uint8_t *const El = L.E + (size_t) blockIdx.x * 64 * (size_t) L.estride + (size_t) lane * 16;      // the lane's first chunk
	auto ebyte = [&](int64_t b) -> uint8_t * { return El + (((b >> 4) << 10) | (b & 15)); };
	const int U = A.U, W1 = A.W1;
	const int dM = A.d[0], dX = A.d[1], dN = A.d[4];
	for(unsigned long long base = (unsigned long long) blockIdx.x * 64; base < L.count; base += (unsigned long long) gridDim.x * 64) {
		const LaneProb X = lt_lane_stage<false>(A, L, base, lane, T, QB);
		LtProb *P = X.P;
		const bool live = X.live;
		const int k = X.k, t_len = X.t_len, q_len = X.q_len, flags = X.flags;
		// (the matrix is indexed by register: no pitch)
		const int low = (t_len + q_len) * (A.MM + U + W1);
		const int off = R - 1 - q_len;                           // register of column 0
		wave_sync();
		// the query codes, eight to a register; the boundary row (nw.c:60-97)
		uint32_t qreg[R / 8], row[R];
#pragma unroll
		for(int w = 0; w < R / 8; ++w) qreg[w] = 0;
#pragma unroll
		for(int j = 0; j < R; ++j) {
			const int n = j - off;
			const bool in = live && n >= 0 && n < q_len;
			const uint32_t code = in ? (uint32_t) QB[n * 64 + lane] : 0u;
			qreg[j >> 3] |= code << ((j & 7) << 2);
			const int D = (in && k != 2) ? W1 + (q_len - 1 - n) * U : 0;
			row[j] = lt_pack16(D, low);
		}
		if(live) {
			for(int n = 0; n <= q_len; ++n) *ebyte(n + off) = (uint8_t) ((n < q_len && k != 2) ? ((n == q_len - 1) ? 18 : 3) : 0);      // row 0
		}
		int score = low, best_m = 0, d0 = 0;
		const int rows = live && !(L.ablate & 4) ? t_len : 0;
		const int rows_max = wave_max(rows);
		uint32_t tw = 0;
		if(rows) tw = T[(((rows - 1) >> 4) << 6) + lane];
		for(int r = 0; r < rows_max; ++r) {
			if(r < rows) {
				const int m = t_len - 1 - r;
				const int tb = (int) ((tw >> (30 - ((m & 15) << 1))) & 3u);
				if((m & 15) == 0 && m) tw = T[(((m - 1) >> 4) << 6) + lane];
				const int bnd = (0 < k) ? 0 : (W1 + r * U);
				const uint32_t bcode = (0 < k) ? 0u : (r ? 5u : 36u);
				// the boundary column (nw.c:100-118), then the row right to left
				int diag = (int) (short) (row[R - 1] & 0xffffu), right = bnd, Qprev = low;
				row[R - 1] = lt_pack16(bnd, low);
				uint32_t ew0 = 0, ew1 = 0, ew2 = 0, ew3 = 0;
				{ const uint32_t b24 = bcode << 24; if((((R - 1) >> 2) & 3) == 3) ew3 = b24; else if((((R - 1) >> 2) & 3) == 2) ew2 = b24; else if((((R - 1) >> 2) & 3) == 1) ew1 = b24; else ew0 = b24; }
				// (what depends on the column only -- a register's query code, whether it is column 0 -- is the same in every row: opaque
				// copies per row keep the compiler from computing all of it in front of the loop, R and 2 R registers that it then spills)
				const int offr = lt_vgpr(off);
#pragma unroll
				for(int w = 0; w < R / 8; ++w) asm volatile("" : "+v"(qreg[w]));
				uint8_t *const er = El + (size_t) (r + 1) * CH * 1024;      // the lane's first chunk of row r + 1
#pragma unroll
				for(int j = R - 2; j >= 0; --j) {
					const uint32_t below = row[j];
					const int Db = (int) (short) (below & 0xffffu), Pb = ((int) below) >> 16;
					const int qb = (int) ((qreg[j >> 3] >> ((j & 7) << 2)) & 15u);
					const bool eq = qb == tb;
					const int sc = eq ? dM : ((qb & 4) ? dN : dX);
					const int Q0 = right + W1, Qe = Qprev + U;
					const bool c1 = Q0 < Qe;
					const int Q = max(Q0, Qe);
					const int P0 = Db + W1, Pe = Pb + U;
					const bool c2 = P0 < Pe;
					const int Pn = max(P0, Pe);
					const int x = diag + sc;
					const int G = max(Pn, Q);
					const int D = max(G, x);
					const bool pw = Pn + (c2 ? 1 : 0) > Q;          // c2 ? Pn >= Q : Pn > Q
					// (both sides of the choice computed, then chosen: left to itself the compiler branches around the gap side, per cell)
					const uint32_t cd = lt_opaque(eq ? 1u : 65u), cg = lt_opaque(pw ? 4u : 2u);
					uint32_t cell = (G <= x) ? cd : cg;
					cell |= (c1 ? 0u : 16u) | (c2 ? 0u : 32u);
					LT_ROW_SET(row[j], lt_pack16(D, Pn));
					d0 = (j == offr) ? D : d0;
					// (the byte made opaque first, or every code constant lives in a register four times, shifted)
					LT_EW_PUT(R, j, lt_opaque(cell));
					diag = Db; right = D; Qprev = Q;
					// (the scheduler would otherwise lift the column-only part of all R cells -- unpacking, query code, P side -- to the
					// front of the row: hundreds of live registers, spilled)
					__builtin_amdgcn_sched_barrier(0);
				}
				if(k < 0 && score < d0) { score = d0; best_m = m; }
			}
		}
		// result selection (nw.c:218-254)
		int sm = 0, sn = 0;
		if(live) {
			if(k < 0) {
				sm = best_m;
				if(k == -2) {
					const int off2 = lt_vgpr(off);          // (not to be compared with R constants in front of the sweep)
#pragma unroll
					for(int j = 0; j < R - 1; ++j) {
						const int Dn = (int) (short) (row[j] & 0xffffu);
						if(j >= off2 && score <= Dn) { score = Dn; sm = 0; sn = j - off2; }
					}
				}
			} else score = d0;
		}
		wave_sync_hbm();   // the matrix is read back by the lane that wrote it
		if(live && !(L.ablate & 2)) {
			RunOut Ro;
			Ro.init(A.runs + P->runs, t_len + q_len + 1);
			int clip = sn, bad = 0;
			// (cell (m, n) lies in row t_len - m, byte n + off: rows run downwards)
			const int q_pos = lt_walk((const uint8_t *) El, -RP, -(t_len - sm), sn + off, 0, sn, (flags & PF_LEAD_TRIM) != 0, &Ro, &clip, (int64_t) RP * (t_len + 1), &bad, 1);
			const int cut = Ro.finish((flags & PF_TRAIL_TRIM) != 0);
			if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
			P->score = score; P->n_runs = bad ? 0 : Ro.n;
			P->clip = (k > 0) ? (q_len - q_pos + cut) : clip;
		}
	}
}

// The same for NW_band (lt_lane_band_kernel's recurrence, nw.c:310-640): the band's cells -- index n = column - (c - band / 2), c the
// band's centre, one column to the left per row -- in R registers, aligned to the RIGHT (index n sits in register n + off, off = R - 2 -
// band: the index beyond the band's last cell is the last register for every lane). The cell below index n is index n - 1 of the row
// before and the diagonal one index n of it, so the row is still updated in place, right to left. What moves from row to row is where a
// row starts (the cell right of the band, or the boundary column once the band reaches the last column: `first`) and where it ends (the
// leftmost cell, which has no template-gap state: `edge`): two comparisons of a register's number with a lane's value per cell, whose
// results replace the cell's values. The query codes of the registers move by one column per row: the code array is shifted by four bits
// (one alignbit per eight registers) and the new code comes from LDS. Registers outside [edge, first] compute garbage that nothing reads;
// their move bytes fall on cells no walk visits, on rows written later, or in front of the matrix (as in lt_reg_kernel).
template <int R>
__global__ __launch_bounds__(64, (R <= 72 ? LT_RB72_WAVES : 2)) void lt_regband_kernel(const LtArgs A, const LaneArgs L) {
	extern __shared__ uint32_t lt_lane_lds[];
	uint32_t *const T = lt_lane_lds + 32;                        // TW x 64 template words
	uint8_t *const QB = (uint8_t *) (T + (size_t) L.TW * 64);   // RQ / 2 x 64 query codes, two columns per byte
	const int lane = threadIdx.x;
	constexpr int RP = (R + 15) & ~15, CH = RP / 16;             // bytes and chunks per row of the interleaved matrix
	uint8_t *const El = L.E + (size_t) blockIdx.x * 64 * (size_t) L.estride + (size_t) lane * 16;      // the lane's first chunk
	auto ebyte = [&](int64_t b) -> uint8_t * { return El + (((b >> 4) << 10) | (b & 15)); };
	const int U = A.U, W1 = A.W1;
	const int dM = A.d[0], dX = A.d[1], dN = A.d[4];
	constexpr int NEG = -(1 << 28);
	constexpr int NW = R / 8;
	static_assert(R % 8 == 0, "eight query codes per register");
	for(unsigned long long base = (unsigned long long) blockIdx.x * 64; base < L.count; base += (unsigned long long) gridDim.x * 64) {
		const LaneProb X = lt_lane_stage<true>(A, L, base, lane, T, QB);
		LtProb *P = X.P;
		const bool live = X.live;
		const int k = X.k, t_len = X.t_len, q_len = X.q_len, flags = X.flags;
		int band = X.band;
		if(band & 1) ++band;
		const int half = band >> 1, bq = band + 1;
		const int low = (t_len + q_len) * (A.MM + U + W1);
		const int off = R - 1 - bq;                              // register of index 0
		wave_sync();
		auto qcode = [&](int col) { col = min(max(col, 0), q_len - 1); return (uint32_t) ((QB[(col >> 1) * 64 + lane] >> ((col & 1) << 2)) & 15u); };
		// the row below the last one (nw.c:386-420): index sn0 is the boundary column; the query codes of the first row's registers
		int c = (t_len + q_len) >> 1;
		const int sn0 = q_len - 1 - (c - half);
		uint32_t qreg[NW], row[R];
#pragma unroll
		for(int w = 0; w < NW; ++w) qreg[w] = 0;
#pragma unroll
		for(int j = 0; j < R; ++j) {
			const int n = j - off;
			qreg[j >> 3] |= (live ? qcode(c - half + n) : 0u) << ((j & 7) << 2);
			int D = 0, Pn = 0;
			if(n >= 0 && n <= sn0) {
				if(n < sn0 && k != 2) D = W1 + (sn0 - n - 1) * U;
				Pn = (n == sn0 && k != 2) ? 0 : low;
			}
			row[j] = lt_pack16(D, Pn);
		}
		if(live) for(int n = 0; n <= sn0; ++n) *ebyte(n + off) = (uint8_t) ((n < sn0 && k != 2) ? ((n == sn0 - 1) ? 18 : 3) : 0);      // row 0
		int en = 0, score = low, bm = 0, bn = 0, d_e = 0;
		const int rows = live && !(L.ablate & 4) ? t_len : 0;
		const int rows_max = wave_max(rows);
		uint32_t tw = 0;
		if(rows) tw = T[(((rows - 1) >> 4) << 6) + lane];
		for(int r = 0; r < rows_max; ++r) {
			if(r < rows) {
				const int m = t_len - 1 - r;
				const int tb = (int) ((tw >> (30 - ((m & 15) << 1))) & 3u);
				if((m & 15) == 0 && m) tw = T[(((m - 1) >> 4) << 6) + lane];
				// where the row starts and ends (open_row of lt_lane_band_kernel)
				int sn;
				bool clipped;
				{
					const int sq = c + half;
					int eq = c - half;
					if(eq < 0) { eq = 0; ++en; } else en = 0;
					if(sq < q_len - 1) { sn = bq - 1; clipped = false; }
					else { sn = en + (q_len - eq) - 1; clipped = true; }
				}
				const int fD = clipped ? ((0 < k) ? 0 : (W1 + r * U)) : low;
				const uint32_t fcode = clipped ? ((0 < k) ? 0u : 37u) : 37u;
				const int fj = lt_vgpr(sn + 1 + off), ej = lt_vgpr(en + off);
				const bool col0 = c - half + en == 0;
				// the code that enters the registers with the next row: column (c - 1) - half - off
				const uint32_t q_next = qcode(c - 1 - half - off);
				const uint32_t tbm = (uint32_t) tb * 0x11111111u;
				uint32_t xq = 0;          // eight query codes ^ template base: 0 = match, bit 2 set = N in the read
				int diag = (int) (short) (row[R - 1] & 0xffffu), right = 0, Qprev = low;
				uint32_t ew0 = 0, ew1 = 0, ew2 = 0, ew3 = 0;
				uint8_t *const er = El + (size_t) (r + 1) * CH * 1024;      // the lane's first chunk of row r + 1
#pragma unroll
				for(int j = R - 1; j >= 0; --j) {
					const uint32_t below = j ? row[j - 1] : 0u;
					const int Dbl = (int) (short) (below & 0xffffu), Pbl = ((int) below) >> 16;
					const bool isf = j == fj, ise = j == ej;
					if((j & 7) == 7) { asm volatile("" : "+v"(qreg[j >> 3])); xq = qreg[j >> 3] ^ tbm; }
					const uint32_t t = (xq >> ((j & 7) << 2)) & 15u;
					const bool eq = t == 0;
					const int sc = eq ? dM : ((t & 4u) ? dN : dX);
					const int Q0 = right + W1, Qe = Qprev + U;
					const bool c1 = Q0 < Qe;
					int Q = max(Q0, Qe);
					const int P0 = Dbl + W1, Pe = Pbl + U;
					const bool c2 = P0 < Pe;
					const int Pm = max(P0, Pe);
					const int PnD = ise ? NEG : Pm;                  // the leftmost cell has no cell below it inside the band
					int PnS = ise ? low : Pm;
					const int x = diag + sc;
					const int G = max(PnD, Q);
					int D = max(G, x);
					const bool pw = PnD + (c2 ? 1 : 0) > Q;          // c2 ? Pn >= Q : Pn > Q
					const uint32_t cd = lt_opaque(eq ? 1u : 65u), cg = lt_opaque(pw ? 4u : 2u);
					uint32_t cell = (G <= x) ? cd : cg;
					cell |= (c1 ? 0u : 16u) | ((c2 || ise) ? 0u : 32u);
					// the cell right of the band, or the boundary column where the band reaches it (nw.c:470-500)
					D = isf ? fD : D; Q = isf ? low : Q; PnS = isf ? low : PnS; cell = isf ? fcode : cell;
					LT_ROW_SET(row[j], lt_pack16(D, PnS));
					d_e = ise ? D : d_e;
					LT_EW_PUT(R, j, lt_opaque(cell));
					diag = Dbl; right = D; Qprev = Q;
					__builtin_amdgcn_sched_barrier(0);
				}
				if(col0 && k < 0 && score < d_e) { score = d_e; bm = m; bn = en; }
				// the next row's columns: every register one column to the left
				--c;
#pragma unroll
				for(int w = NW - 1; w > 0; --w) qreg[w] = (qreg[w] << 4) | (qreg[w - 1] >> 28);
				qreg[0] = (qreg[0] << 4) | q_next;
			}
		}
		// result selection (nw.c:557-585): the leftmost cell of the first row unless a row above scored higher in column 0
		int q_pos = 0;
		if(live) {
			if(bm == 0) { bn = en; score = d_e; }
			if(k == -2) {
				const int e2 = lt_vgpr(en + off);
#pragma unroll
				for(int j = 0; j < R - 1; ++j) {
					const int Dn = (int) (short) (row[j] & 0xffffu);
					if(j >= e2 && score <= Dn) { score = Dn; bm = 0; bn = j - (e2 - en); q_pos = j - e2; }
				}
			}
		}
		wave_sync_hbm();
		if(live && !(L.ablate & 2)) {
			RunOut Ro;
			Ro.init(A.runs + P->runs, t_len + q_len + 1);
			int clip = q_pos, bad = 0;
			// (band cell (m, n) lies in row t_len - m, byte n + off: a diagonal step keeps the byte, rows run downwards)
			const int qend = lt_walk((const uint8_t *) El, -RP, -(t_len - bm), bn + off, -1, q_pos, (flags & PF_LEAD_TRIM) != 0, &Ro, &clip, (int64_t) RP * (t_len + 1), &bad, 1);
			const int cut = Ro.finish((flags & PF_TRAIL_TRIM) != 0);
			if(bad) atomicMax(&A.counters[LCI(LC_STATUS)], 10ull);
			P->score = score; P->n_runs = bad ? 0 : Ro.n;
			P->clip = (k > 0) ? (q_len - qend + cut) : clip;
		}
	}
}

// ---- per read: the runs of its problems and MEMs in chain order, merged; alignment figures; the read filter of assemble_KMA
// (assembly.c:1931-1961: + Wl for an alignment that starts at the first / ends at the last template base, minlen, mrc, scoreT)
__global__ __launch_bounds__(64) void lt_finish_kernel(const LtArgs A) {
	const int lane = threadIdx.x;
	uint32_t *const tmp = A.tmp + (size_t) blockIdx.x * A.tmp_cap;
	for(int64_t rr = blockIdx.x; rr < A.n_reads; rr += gridDim.x) {
		const int64_t r = A.r0 + rr;
		const LtRead H = A.rd[rr];
		if(lane == 0) {
			for(int x = 0; x < 10; ++x) A.o_stats[10 * r + x] = 0;
			A.o_off[r] = 0; A.o_nops[r] = 0;
			if(A.o_rc) A.o_rc[r] = H.rc;
		}
		if(!H.status) { if(A.score_mode && lane == 0) A.o_stats[10 * r + 3] = 1; continue; }          // (KMA_score's failure value: len 1)
		const int q_len = A.len[r];
		const int tt = A.tmpl ? A.tmpl[r] : A.tmpl_all;
		const int t_len = A.db.tlen[abs(tt)];
		// ---- phase A: every problem's runs (+ the MEM behind it) into one list ----
		int n_ent = 0, score = 0, lead_cols = 0, lead_clip = 0, trail_clip = 0;
		bool over = false;
		for(int b = 0; b < H.n_prob; b += 64) {
			const int l = b + lane;
			int cnt = 0, fl = PF_NONE, nr = 0, body = 0, deg_len = 0, sc = 0;
			int64_t src = 0;
			if(l < H.n_prob) {
				const LtProb *P = A.prob + H.first + l;
				fl = P->flags; body = P->body; src = P->runs;
				sc = P->body_score;
				if(fl & PF_NONE) nr = 0;
				else if(fl & (PF_DEGEN_I | PF_DEGEN_D)) { nr = 1; deg_len = (fl & PF_DEGEN_I) ? P->q_l : P->t_l; sc += A.W1 + (deg_len - 1) * A.U; }
				else {
					nr = P->n_runs; sc += P->score;
					if(l == 0) lead_clip = P->clip;
					if(l == H.n_prob - 1) trail_clip = P->clip;
				}
				cnt = nr + (body > 0 ? 1 : 0);
			}
			const int incl = wave_scan_incl(cnt, lane);
			const int tot = __shfl(incl, 63);
			int o = n_ent + incl - cnt;
			if((int64_t) n_ent + tot > A.tmp_cap) { over = true; break; }
			if(fl & (PF_DEGEN_I | PF_DEGEN_D)) tmp[o++] = ((uint32_t) deg_len << 2) | ((fl & PF_DEGEN_I) ? 2u : 3u);
			else for(int x = 0; x < nr; ++x) {
				const uint32_t e = A.runs[src + x];
				tmp[o++] = e;
				if(l == 0 && (e & 3u) != 2u) lead_cols += (int) (e >> 2);
			}
			if(body > 0) tmp[o++] = (uint32_t) body << 2;
			score += sc;
			n_ent += tot;
		}
		if(over) { if(lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 9ull); continue; }
		score = wave_sum(score);
		lead_cols = __shfl(lead_cols, 0); lead_clip = __shfl(lead_clip, 0);
		trail_clip = __shfl(trail_clip, (H.n_prob - 1) & 63);
		wave_sync_hbm();
		// ---- phase B: neighbours of one class merge; two passes (count, then write) ----
		int n_merged = 0, cols[4] = {0, 0, 0, 0};
		for(int c0 = 0; c0 < n_ent; c0 += 64) {
			const bool valid = c0 + lane < n_ent;
			const uint32_t e = valid ? tmp[c0 + lane] : 0u;
			int pc = __shfl_up((int) (e & 3u), 1);
			if(lane == 0) pc = c0 ? (int) (tmp[c0 - 1] & 3u) : -1;
			n_merged += __popcll(__ballot(valid && (int) (e & 3u) != pc));
			if(valid) cols[e & 3u] += (int) (e >> 2);
		}
		const int match = wave_sum(cols[0] + cols[1]), tGaps = wave_sum(cols[2]), qGaps = wave_sum(cols[3]);
		const int aln_len = match + tGaps + qGaps;
		// the filter
		const int start = H.pos0 - lead_cols;
		if(A.score_mode) {
			// stage 3a: KMA_score's figures as they are (the caller applies alnFragsSE's filter)
			if(lane == 0) { int32_t *st = A.o_stats + 10 * r; st[0] = score; st[1] = start; st[3] = aln_len; st[6] = match; st[7] = tGaps; st[8] = qGaps; st[9] = H.mapQ; }
			continue;
		}
		int end = start + aln_len - tGaps;
		if(t_len < end) end -= t_len;
		int read_score = score;
		if(start == 0) read_score += A.Wl;
		if(end == t_len) read_score += A.Wl;
		double norm = 0;
		if(A.minlen <= aln_len && ((A.mrc * q_len <= aln_len - qGaps) || (A.mrc * t_len <= aln_len - tGaps))) norm = 1.0 * read_score / aln_len;
		else read_score = 0;
		if(!(0 < read_score && A.scoreT <= norm)) continue;
		unsigned long long ob = 0;
		if(lane == 0) ob = atomicAdd(A.ops_top, (unsigned long long) n_merged);
		ob = __shfl(ob, 0);
		if((int64_t) ob + n_merged > A.ops_cap) { if(lane == 0) atomicMax(&A.counters[LCI(LC_STATUS)], 2ull); continue; }
		uint32_t *out = A.ops + ob;
		int w = 0, carry_cls = -1, carry_len = 0;
		for(int c0 = 0; c0 < n_ent; c0 += 64) {
			const bool valid = c0 + lane < n_ent;
			const uint32_t e = valid ? tmp[c0 + lane] : 0u;
			const int cls = (int) (e & 3u), len = valid ? (int) (e >> 2) : 0;
			int pc = __shfl_up(cls, 1);
			if(lane == 0) pc = carry_cls;
			const bool st = valid && cls != pc;
			const unsigned long long sm = __ballot(st);
			const int incl = wave_scan_incl(len, lane);
			const int before = incl - len;                       // columns of this round in front of the lane
			// a lane that opens a run closes the one before it (the shuffles are executed by all lanes)
			const unsigned long long lower = sm & ((1ull << lane) - 1ull);
			const int a = lower ? 63 - __clzll((long long) lower) : 0;
			const int before_a = __shfl(before, a);
			const int cls_a = __shfl(cls, a);
			if(st) {
				const int j = __popcll(lower);
				if(lower) out[w + j - (carry_cls < 0 ? 1 : 0)] = ((uint32_t) (before - before_a) << 2) | (uint32_t) cls_a;
				else if(carry_cls >= 0) out[w] = ((uint32_t) (carry_len + before) << 2) | (uint32_t) carry_cls;
			}
			const int total = __shfl(incl, 63);
			if(sm) {
				const int last = 63 - __clzll((long long) sm);
				w += __popcll(sm) - (carry_cls < 0 ? 1 : 0);
				carry_cls = __shfl(cls, last);
				carry_len = total - __shfl(before, last);
			} else carry_len += total;
		}
		if(lane == 0 && carry_cls >= 0) out[w] = ((uint32_t) carry_len << 2) | (uint32_t) carry_cls;
		if(lane == 0) {
			int32_t *st = A.o_stats + 10 * r;
			st[0] = read_score; st[1] = start; st[2] = (t_len < end) ? end - t_len : end; st[3] = aln_len;
			st[4] = H.clip0 + lead_clip; st[5] = q_len - H.qe_trail + trail_clip;
			st[6] = match; st[7] = tGaps; st[8] = qGaps; st[9] = H.mapQ;
			A.o_off[r] = (int64_t) ob; A.o_nops[r] = n_merged;
		}
	}
}

} // namespace

// Streams that run side by side. Which hardware queue a stream lands on is the runtime's business (the first few streams of a process
// get queues of their own, later ones share, by rules that changed between releases), so it is measured: two short spinning kernels, one
// wavefront each, on two streams take one spin's time when the streams sit on different queues and two when they share one. Out of
// eight candidates the first `want` that are pairwise side by side are kept (a few dozen launches of 0.2 ms, once per process).
namespace {
__global__ void lt_spin_kernel(long long ticks) {
	const long long t0 = wall_clock64();
	while(wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}
static int lt_pick_streams(hipStream_t *out, int want) {
	constexpr int NC = 8;
	hipStream_t cand[NC];
	for(int i = 0; i < NC; ++i) HIP_TRY(hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking));
	const long long ticks = 20000;          // 0.2 ms of the 100 MHz wall clock
	auto pair_ms = [&](hipStream_t a, hipStream_t b) -> double {
		(void) hipStreamSynchronize(a); (void) hipStreamSynchronize(b);
		const auto t0 = std::chrono::steady_clock::now();
		hipLaunchKernelGGL(lt_spin_kernel, dim3(1), dim3(64), 0, a, ticks);
		hipLaunchKernelGGL(lt_spin_kernel, dim3(1), dim3(64), 0, b, ticks);
		(void) hipStreamSynchronize(a); (void) hipStreamSynchronize(b);
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	};
	(void) pair_ms(cand[0], cand[1]);          // (first launches)
	int sel[NC], ns = 0;
	bool used[NC] = {false};
	sel[ns++] = 0; used[0] = true;
	for(int c = 1; c < NC && ns < want; ++c) {
		bool ok = true;
		for(int x = 0; x < ns && ok; ++x) {
			// the shorter of two tries: a try that something else delayed must not rule a stream out
			const double ms = std::min(pair_ms(cand[sel[x]], cand[c]), pair_ms(cand[sel[x]], cand[c]));
			ok = ms < 0.34;
		}
		if(ok) { sel[ns++] = c; used[c] = true; }
	}
	for(int c = 1; c < NC && ns < want; ++c) if(!used[c]) { sel[ns++] = c; used[c] = true; }      // (fewer queues than wanted: any)
	for(int x = 0; x < want; ++x) out[x] = cand[sel[x]];
	for(int c = 0; c < NC; ++c) if(!used[c]) (void) hipStreamDestroy(cand[c]);
	if(getenv("KMAHIP_DEBUG_TIMING") || getenv("KMAHIP_LT_STREAM_REPORT")) { fprintf(stderr, "[kmahip] longtrace: streams"); for(int x = 0; x < want; ++x) fprintf(stderr, " %d", sel[x]); fprintf(stderr, " of %d candidates\n", NC); }
	return KMAHIP_OK;
}

// the process's worker streams (up to four), picked once: for every pipeline that runs kernels side by side (declared where it is used)
int kmahip_worker_streams(hipStream_t *out, int want) {
	static hipStream_t wk[4] = {nullptr, nullptr, nullptr, nullptr};
	if(!wk[0]) { const int rc = lt_pick_streams(wk, 4); if(rc) return rc; }
	for(int x = 0; x < want && x < 4; ++x) out[x] = wk[x];
	return KMAHIP_OK;
}

static int lt_reserve(kmahip_ws *ws, int slot, size_t bytes) {
	if(ws->lt_bytes[slot] >= bytes) return KMAHIP_OK;
	(void) hipFree(ws->lt_buf[slot]);
	ws->lt_buf[slot] = nullptr; ws->lt_bytes[slot] = 0;
	// (an eighth more than asked for: what is sized by a pass's own counts -- the lanes' matrices, the sort's scratch -- must not be
	// freed and allocated again, gigabytes at a time, because the next pass needs a few bytes more)
	const size_t want = bytes + bytes / 8;
	if(hipMalloc(&ws->lt_buf[slot], want) == hipSuccess) { ws->lt_bytes[slot] = want; return KMAHIP_OK; }
	if(hipMalloc(&ws->lt_buf[slot], bytes) != hipSuccess) { kmahip_set_error("hipMalloc of %zu bytes (long-read trace scratch) failed", bytes); return KMAHIP_ENOMEM; }
	ws->lt_bytes[slot] = bytes;
	return KMAHIP_OK;
}

// The pipeline over a device-resident batch, in passes of about 4e8 bases (the pools of a pass: one descriptor per chain join,
// one run slot word per DP column at most). tmpl == NULL: every read against tmpl_all. rc_in == NULL: both strands are seeded
// and anker_rc decides (`-Mt1`), else the orientation is given. rc_out (may be NULL): the strand that was aligned.
int kmahip_launch_longtrace(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *tmpl, int tmpl_all, const int32_t *rc_in,
                            const uint8_t *tmpl_ok, int one2one, const kmahip_params *p, kmahip_traces *out, int32_t *rc_out, hipStream_t stream, int score_mode) {
	const int64_t n = reads->n_reads;
	if(n < 0 || !p || !out || !out->stats || !out->ops_off || !out->n_ops || (out->ops_cap > 0 && !out->ops)) { kmahip_set_error("bad arguments"); return KMAHIP_EINVAL; }
	if(!db->dev.tpos_slots) { kmahip_set_error("index has no .length.b/.seq.b: stage 3c unavailable"); return KMAHIP_EINVAL; }
	if(!tmpl && (tmpl_all < 1 || (uint32_t) tmpl_all >= db->info.DB_size)) { kmahip_set_error("template %d out of range", tmpl_all); return KMAHIP_EINVAL; }
	if(n == 0) return KMAHIP_OK;
	const int max_len = reads->max_len;
	if(max_len <= 0 || max_len > (1 << 24)) { kmahip_set_error("kmahip_reads.max_len must be set (<= 2^24) for the trace stage"); return KMAHIP_EINVAL; }
	// (the seeding kernel waits for its index lookups: as many wavefronts as the registers allow, 20 per CU)
	const int seed_wgs = (int) std::min<int64_t>(getenv("KMAHIP_LT_SEED_WGS") ? atoi(getenv("KMAHIP_LT_SEED_WGS")) : 5120, std::max<int64_t>(1024, 400000000ll / max_len));
	const int fin_wgs = 2048, dp_wgs = 2048, dpx_wgs = 1024;
	int mcap = std::max(1024, max_len / 8 + 256);          // (MEM slots per seeding wavefront: four times more whenever a read runs out, below)
	const int64_t tmp_cap = 4ll * max_len + 1024;
	const int64_t xe_cap = 2ll << 20;
	const int xrow = max_len + 72;
	// lane classes (lt_lane_kernel): as long as the scores of (rows + columns) cells fit 16 bits; KMAHIP_LT_LANE=0 switches them off
	int lane_tq = 0;
	{
		int mx = std::max(std::max(abs(p->rw.M), abs(p->rw.MM)), std::max(abs(p->rw.U), abs(p->rw.W1)));
		for(int i = 0; i < 25; ++i) mx = std::max(mx, abs(p->rw.d[i / 5][i % 5]));
		const char *e = getenv("KMAHIP_LT_LANE");
		if(!(e && e[0] == '0')) lane_tq = 32000 / (abs(p->rw.MM + p->rw.U + p->rw.W1) + 2 * mx + 1);
	}
	// three scores only (match, mismatch, N in the read)? Then the lane kernels compute the score of a pair
	bool simple_sc = true;
	for(int i = 0; i < 4; ++i) for(int j = 0; j < 5; ++j) {
		const int want = j == 4 ? p->rw.d[0][4] : (i == j ? p->rw.d[0][0] : p->rw.d[0][1]);
		if(p->rw.d[i][j] != want) simple_sc = false;
	}
	if(getenv("KMAHIP_LT_SCORE_TABLE")) simple_sc = false;
	const bool reg_rows = simple_sc && !(getenv("KMAHIP_LT_REG") && getenv("KMAHIP_LT_REG")[0] == '0');      // lt_reg_kernel for the classes of up to 64 cells a row
	const bool reg_wide = reg_rows && !(getenv("KMAHIP_LT_REG128") && getenv("KMAHIP_LT_REG128")[0] == '0');      // ... and of up to 128 (one wavefront per SIMD: 512 registers)
	const bool reg_band = reg_rows && !(getenv("KMAHIP_LT_REGBAND") && getenv("KMAHIP_LT_REGBAND")[0] == '0');  // lt_regband_kernel for the banded classes of up to 96 cells a row
	struct LaneLaunch { LaneGeom g; int wgs; size_t lds; size_t e_off; size_t estride; };      // estride: bytes of move matrix per lane
	LaneLaunch lg[LT_LCLS];
	for(int j = 0; j < LT_LCLS; ++j) {
		lg[j].g = lt_lane_geom(j);
		lg[j].lds = (size_t) (32 + lg[j].g.R * 64 + lg[j].g.TW * 64) * 4 + (size_t) lg[j].g.RQ * (j < LT_LFULL ? 64 : 32);
		lg[j].wgs = 256 * (int) std::min<size_t>(16, (160 * 1024) / lg[j].lds);
		// (lt_reg_kernel keeps no row in LDS: its wavefronts per CU follow from its registers -- 5 / 5 / 4 / 3 per SIMD)
		if(j <= 3 && reg_rows) lg[j].wgs = 256 * 4 * (j == 0 ? LT_REG16_WAVES : j == 1 ? LT_REG32_WAVES : j == 2 ? LT_REG48_WAVES : LT_REG64_WAVES);
		if(j == 4 && reg_wide) lg[j].wgs = 256 * 4;
		// (lt_regband_kernel likewise: 3 / 2 per SIMD)
		if((j == LT_LFULL || j == LT_LFULL + 1) && reg_band) lg[j].wgs = 256 * (j == LT_LFULL ? 4 * LT_RB72_WAVES : 8);
		lg[j].e_off = 0;
		lg[j].estride = (size_t) lg[j].g.ecap + 288;
	}
	// (the register kernels' matrices: rows of R rounded up to 16 bytes, one per template row the class takes + the boundary row)
	for(int j = 0; j < LT_LCLS; ++j) {
		const bool regk = (j <= 3 && reg_rows) || (j == 4 && reg_wide) || ((j == LT_LFULL || j == LT_LFULL + 1) && reg_band);
		if(regk) lg[j].estride = (size_t) ((lg[j].g.R + 15) & ~15) * (size_t) (16 * lg[j].g.TW + 1) + 32;
	}
	int64_t B = std::min<int64_t>(n, std::max<int64_t>(1024, 400000000ll / max_len));
	if(const char *e = getenv("KMAHIP_LT_PASS_READS")) B = std::min<int64_t>(n, std::max<int64_t>(1, atoll(e)));          // (the tests: many passes out of a few reads)
	int64_t prob_cap = B * (max_len / 16 + 4), runs_cap = B * (3ll * max_len + 64);
	int rc;
	if((rc = lt_reserve(ws, 0, (size_t) seed_wgs * 7 * mcap * 4)) || (rc = lt_reserve(ws, 5, (size_t) fin_wgs * tmp_cap * 4)) ||
	   (rc = lt_reserve(ws, 6, (size_t) dpx_wgs * ((size_t) xe_cap + 16 * (size_t) xrow))) || (rc = lt_reserve(ws, 7, LCI(LC_N + 1) * 8))) return rc;
	unsigned long long *counters = (unsigned long long *) ws->lt_buf[7];
	HIP_TRY(hipMemsetAsync(counters, 0, LCI(LC_N + 1) * 8, stream));
	for(int i = 0; i < 4; ++i) ws->lt_stats[i] = 0;
	LtArgs A;
	A.db = db->dev;
	A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.tmpl = tmpl; A.rc_in = rc_in; A.tmpl_ok = tmpl_ok; A.tmpl_all = tmpl_all; A.one2one = one2one; A.exhaustive = p->exhaustive;
	A.score_mode = score_mode;
	A.q_start = rc_in ? reads->q_start : nullptr; A.q_end = rc_in ? reads->q_end : nullptr;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1; A.Wl = p->rw.Wl;
	for(int i = 0; i < 25; ++i) A.d[i] = p->rw.d[i / 5][i % 5];
	A.minlen = p->minlen; A.mq = p->mq; A.scoreT = p->scoreT; A.mrc = p->mrc; A.ts = score_mode ? 0 : p->ts;
	A.mem = (int32_t *) ws->lt_buf[0]; A.mcap = mcap;
	A.tmp = (uint32_t *) ws->lt_buf[5]; A.tmp_cap = tmp_cap;
	A.xE = (uint8_t *) ws->lt_buf[6]; A.xe_cap = xe_cap; A.xrow = xrow;
	A.counters = counters;
	A.o_stats = out->stats; A.o_off = out->ops_off; A.o_nops = out->n_ops; A.ops = out->ops; A.ops_cap = out->ops_cap;
	A.ops_top = counters + LCI(LC_OUT); A.o_rc = rc_out;
	A.rec = nullptr;
	A.stop = getenv("KMAHIP_LT_STOP") ? atoi(getenv("KMAHIP_LT_STOP")) : 0;
	if(getenv("KMAHIP_DEBUG_TIMING")) {
		static uint32_t *rec = nullptr;
		if(!rec && hipHostMalloc((void **) &rec, 2048 * 16 * 4, hipHostMallocMapped) != hipSuccess) rec = nullptr;
		if(rec) { memset(rec, 0, 2048 * 16 * 4); uint32_t *drec = nullptr; if(hipHostGetDevicePointer((void **) &drec, rec, 0) == hipSuccess) A.rec = drec; }
	}
	// Two sets of a pass's pools and counters: the seeding of pass i + 1 (index lookups: it waits) runs on a stream of its own beside the
	// DP kernels of pass i (KMAHIP_LT_PIPE=0, one pass, or KMAHIP_DEBUG_TIMING: one set, one after the other). The run pool's top
	// (ops_top) is one word for all passes, in the first set's counters.
	const bool dbg0 = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	const bool piped = !dbg0 && n > B && !(getenv("KMAHIP_LT_PIPE") && getenv("KMAHIP_LT_PIPE")[0] == '0');
	static const int SLOT[2][6] = {{1, 2, 3, 4, 8, 7}, {11, 12, 13, 14, 15, 16}};          // rd, prob, runs, queue, lane queue + keys, counters
	// pinned host memory of the workspace (slot 17 of its buffers; api.hip frees it as such): the counters after the seeding ([x]) and
	// after the finish ([2 + x]) of a set
	if(!ws->lt_buf[17]) {
		if(hipHostMalloc(&ws->lt_buf[17], (size_t) 4 * LCI(LC_N) * 8, hipHostMallocDefault) != hipSuccess) { ws->lt_buf[17] = nullptr; kmahip_set_error("hipHostMalloc failed"); return KMAHIP_ENOMEM; }
		ws->lt_bytes[17] = (size_t) 4 * LCI(LC_N) * 8;
	}
	unsigned long long *const hc = (unsigned long long *) ws->lt_buf[17];
	// FOUR streams of the pipeline's own that were SEEN to run side by side (lt_pick_streams): the runtime maps streams onto a few
	// hardware queues, and streams that share a queue run their kernels one after the other. With the caller's stream as the first of the
	// four and three more made on first use, the streams a process had made before decided whether the seeding shared a queue with the
	// sweeps (tools/lt_streams_exp.py: 176 ms for the stage at 200 k reads with 0 or 4 streams made before, 192-204 with 1-3: the bench's C4
	// leg, behind its other legs, against the same leg alone). The caller's stream waits for the work; it does none of it.
	constexpr int NSIDE = 3;
	hipStream_t wk[NSIDE + 1];
	{ const int rcw = kmahip_worker_streams(wk, NSIDE + 1); if(rcw) return rcw; }
	hipStream_t *const side = wk + 1;
	const hipStream_t caller = stream;
	if(!dbg0) {
		hipEvent_t e0 = nullptr;
		HIP_TRY(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
		HIP_TRY(hipEventRecord(e0, caller));
		HIP_TRY(hipStreamWaitEvent(wk[0], e0, 0));
		(void) hipEventDestroy(e0);
		stream = wk[0];
	}
	struct Evs { hipEvent_t seed[2] = {nullptr, nullptr}, fin[2] = {nullptr, nullptr}; ~Evs() { for(int x = 0; x < 2; ++x) { if(seed[x]) (void) hipEventDestroy(seed[x]); if(fin[x]) (void) hipEventDestroy(fin[x]); } } } ev;
	for(int x = 0; x < 2; ++x) { HIP_TRY(hipEventCreateWithFlags(&ev.seed[x], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ev.fin[x], hipEventDisableTiming)); }
	bool fin_pending[2] = {false, false};
	int64_t fin_nb[2] = {0, 0};
	LtArgs Ap[2];
	// the pools of set x at their present sizes, and A pointed at them for reads [r0, r0 + nb)
	auto set_args = [&](int x, int64_t r0, int64_t nb) -> int {
		const int *S = SLOT[x];
		int rc2;
		if((rc2 = lt_reserve(ws, S[0], (size_t) B * sizeof(LtRead))) || (rc2 = lt_reserve(ws, S[1], (size_t) prob_cap * sizeof(LtProb))) ||
		   (rc2 = lt_reserve(ws, S[2], (size_t) runs_cap * 4)) || (rc2 = lt_reserve(ws, S[3], (size_t) LT_NCLS * prob_cap * 4)) ||
		   (lane_tq && (rc2 = lt_reserve(ws, S[4], (size_t) 4 * prob_cap * 4))) || (rc2 = lt_reserve(ws, S[5], LCI(LC_N + 1) * 8))) return rc2;
		A.lq = (int32_t *) ws->lt_buf[S[4]]; A.lkey = lane_tq ? (uint32_t *) ws->lt_buf[S[4]] + prob_cap : nullptr; A.lane_tq = lane_tq; A.lane_mask = getenv("KMAHIP_LT_LANE") && getenv("KMAHIP_LT_LANE")[0] == 'f' ? 1 : (getenv("KMAHIP_LT_LANE") && getenv("KMAHIP_LT_LANE")[0] == 'b' ? 2 : 3);
		// a lane takes problems of up to this many cells: the longest problem of a class is what its kernel lasts (tools/c4_time.py, 200 k
		// reads, trace stage: no cap 264 ms, 24 000 234, 16 000 226, 10 000 259, 6 000 409; with the passes overlapped 13 000 214, 16 000 210,
		// 20 000 204, 24 000 205)
		A.lane_turns = getenv("KMAHIP_LT_LANE_TURNS") ? atoi(getenv("KMAHIP_LT_LANE_TURNS")) : 20000;
		// bits 8 + j: lane class j in use (KMAHIP_LT_LCLS: a bit per class). Not class 8 (bands of over 93): a few thousand problems a pass, a
		// hundred wavefronts that last as long as their longest problem -- the wavefront-per-problem kernels take them in a tenth of the time
		A.lane_mask |= (getenv("KMAHIP_LT_LCLS") ? (int) strtol(getenv("KMAHIP_LT_LCLS"), nullptr, 0) & 0x1ff : 0x0ff) << 8;
		A.r0 = r0; A.n_reads = nb;
		A.rd = (LtRead *) ws->lt_buf[S[0]]; A.prob = (LtProb *) ws->lt_buf[S[1]]; A.prob_cap = prob_cap;
		A.runs = (uint32_t *) ws->lt_buf[S[2]]; A.runs_cap = runs_cap; A.queue = (int32_t *) ws->lt_buf[S[3]];
		A.counters = (unsigned long long *) ws->lt_buf[S[5]];
		A.ops_top = (unsigned long long *) ws->lt_buf[7] + LCI(LC_OUT);
		A.mem = (int32_t *) ws->lt_buf[0]; A.mcap = mcap;
		return KMAHIP_OK;
	};
	auto launch_seed = [&](int x, hipStream_t s) -> int {
		HIP_TRY(hipMemsetAsync(A.counters, 0, LCI(LC_OUT) * 8, s));
		if(dbg0) { fprintf(stderr, "[kmahip] longtrace pass %lld+%lld: seeding (prob_cap %lld, runs_cap %lld)\n", (long long) A.r0, (long long) A.n_reads, (long long) prob_cap, (long long) runs_cap); fflush(stderr); }
		hipLaunchKernelGGL(lt_seed_kernel, dim3((unsigned) std::min<int64_t>(seed_wgs, A.n_reads)), dim3(64), 0, s, A);
		HIP_TRY(hipMemcpyAsync(hc + (size_t) x * LCI(LC_N), A.counters, (size_t) LCI(LC_N) * 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipEventRecord(ev.seed[x], s));
		Ap[x] = A;
		return KMAHIP_OK;
	};
	// what the finish of set x's last pass left: status, work figures
	auto fin_check = [&](int x) -> int {
		if(!fin_pending[x]) return KMAHIP_OK;
		fin_pending[x] = false;
		HIP_TRY(hipEventSynchronize(ev.fin[x]));
		const unsigned long long *cf = hc + (size_t) (2 + x) * LCI(LC_N);
		if(cf[LCI(LC_STATUS)] == 10) { kmahip_set_error("long-read trace: a move matrix was left through a non-boundary cell (internal error)"); return KMAHIP_EDEVICE; }
		ws->lt_stats[0] += cf[LCI(LC_PROB)]; ws->lt_stats[1] += cf[LCI(LC_CELLS)]; ws->lt_stats[2] += cf[LCI(LC_MEMS)]; ws->lt_stats[3] += (unsigned long long) fin_nb[x];
		if(cf[LCI(LC_STATUS)] == 8 || cf[LCI(LC_STATUS)] == 9) { kmahip_set_error("long-read trace: a DP problem or a read's run list beyond the scratch (status %llu)", cf[LCI(LC_STATUS)]); return KMAHIP_EDEVICE; }
		return KMAHIP_OK;
	};
	int px = 0;              // the set of the pass in hand
	bool seeded = false;     // its seeding was started beside the pass before
	for(int64_t r0 = 0; r0 < n;) {
		const int64_t nb = std::min<int64_t>(B, n - r0);
		if(!seeded) {
			if((rc = fin_check(px)) || (rc = set_args(px, r0, nb)) || (rc = launch_seed(px, stream))) return rc;
		}
		seeded = false;
		A = Ap[px];
		HIP_TRY(hipEventSynchronize(ev.seed[px]));
		const unsigned long long *c = hc + (size_t) px * LCI(LC_N);
		if(c[LCI(LC_STATUS)] == 5 || c[LCI(LC_STATUS)] == 6) {
			// a pool of the pass ran out: larger pools (or a smaller pass), same reads again
			if(c[LCI(LC_STATUS)] == 5) prob_cap = std::max<int64_t>(2 * prob_cap, (int64_t) c[LCI(LC_PROB)] + 1024);
			else runs_cap = std::max<int64_t>(2 * runs_cap, (int64_t) c[LCI(LC_RUNS)] + 1024);
			if(prob_cap * (int64_t) sizeof(LtProb) > (48ll << 30) || runs_cap * 4 > (96ll << 30)) { kmahip_set_error("long-read trace: pools of a pass beyond 96 GB"); return KMAHIP_ENOMEM; }
			continue;
		}
		if(c[LCI(LC_STATUS)] == 3) {
			// a read full of repeats: more MEMs against its template than a wavefront has slots for -- more slots, same reads again
			if(mcap >= (1 << 16)) { kmahip_set_error("seed (MEM) capacity per read exceeded (%d MEMs)", mcap); return KMAHIP_EOVERFLOW; }
			mcap *= 4;
			if((rc = lt_reserve(ws, 0, (size_t) seed_wgs * 7 * mcap * 4))) return rc;
			A.mem = (int32_t *) ws->lt_buf[0]; A.mcap = mcap;
			if(getenv("KMAHIP_DEBUG_TIMING")) fprintf(stderr, "[kmahip] longtrace: seed (MEM) capacity per read raised to %d\n", mcap);
			continue;
		}
		const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
		auto stage = [&](const char *what) {
			if(!dbg) return;
			if(A.rec) {
				// a kernel that does not come back within 15 s: dump the recorder and give up (the process must not sit on a hung GPU)
				for(int us = 0; hipStreamQuery(stream) == hipErrorNotReady; us += 100) {
					struct timespec ts = {0, 100000};
					nanosleep(&ts, nullptr);
					if(us < 4000000) continue;
					fprintf(stderr, "[kmahip] longtrace: %s does not finish; recorder of the workgroups still busy:\n", what);
					int shown = 0, hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
					for(int wg = 0; wg < 1024; ++wg) { const volatile uint32_t *rr = A.rec + (size_t) wg * 16; hist[rr[1] == 99 ? 6 : (rr[1] < 6 ? rr[1] : 7)]++; }
					fprintf(stderr, "  phase histogram (0 idle, 1 got problem, 2 coop, 3 serial, 4 swept, 5 walked, 99 exited, other): %d %d %d %d %d %d %d %d\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
					for(int wg = 0; wg < 2048 && shown < 12; ++wg) {
						const volatile uint32_t *rr = A.rec + (size_t) wg * 16;
						if(rr[1] == 99 || (rr[1] == 0 && rr[11] == 0)) continue;
						fprintf(stderr, "  wg %d:", wg);
						for(int x = 0; x < 16; ++x) fprintf(stderr, " %d", (int) rr[x]);
						fprintf(stderr, "\n");
						++shown;
					}
					fflush(stderr);
					_exit(3);
				}
			}
			const hipError_t e = hipStreamSynchronize(stream);
			static auto t_last = std::chrono::steady_clock::now();
			const auto t_now = std::chrono::steady_clock::now();
			fprintf(stderr, "[kmahip] longtrace pass %lld+%lld: %s done (%s), %.2f ms since the stage before\n", (long long) r0, (long long) nb, what, hipGetErrorString(e),
			        std::chrono::duration<double, std::milli>(t_now - t_last).count());
			t_last = t_now;
			fflush(stderr);
		};
		stage("seed");
		if(dbg) { fprintf(stderr, "[kmahip] longtrace: seeded; status %llu, %llu problems, %llu run words, classes %llu %llu %llu %llu | %llu %llu %llu %llu | %llu | %llu %llu %llu %llu | %llu %llu %llu %llu\n", c[LCI(LC_STATUS)], c[LCI(LC_PROB)], c[LCI(LC_RUNS)],
		                  c[LCI(LC_CNT)], c[LCI(LC_CNT + 1)], c[LCI(LC_CNT + 2)], c[LCI(LC_CNT + 3)], c[LCI(LC_CNT + 4)], c[LCI(LC_CNT + 5)], c[LCI(LC_CNT + 6)], c[LCI(LC_CNT + 7)], c[LCI(LC_CNT + 8)],
		                  c[LCI(LC_CNT + 9)], c[LCI(LC_CNT + 10)], c[LCI(LC_CNT + 11)], c[LCI(LC_CNT + 12)], c[LCI(LC_CNT + 13)], c[LCI(LC_CNT + 14)], c[LCI(LC_CNT + 15)], c[LCI(LC_CNT + 16)]); fflush(stderr); }
		// grids follow the queue lengths: 4 wavefronts per workgroup of lt_dp_kernel, 64 / W problems per wavefront round
		auto wgs = [&](int cls, int per_wg, int cap) { return dim3((unsigned) std::min<unsigned long long>((unsigned long long) cap, (c[LCI(LC_CNT + cls)] + per_wg - 1) / per_wg)); };
		// The size classes are independent of each other: the sweeps that live in LDS run side by side on three streams, the
		// kernels that share the per-workgroup HBM scratch (classes 9-12 and the one-lane class 8: a few hundred big problems that
		// take as long as the rest together when they run alone) one after the other on a fourth. With KMAHIP_DEBUG_TIMING
		// everything stays on the caller's stream so that the stages can be timed.
		int32_t *vals_out = nullptr;
		if(c[LCI(LC_LANE)]) {
			// the lane classes: one sort by (class, sweep length), longest first, then a kernel per class (below)
			const size_t nl = (size_t) c[LCI(LC_LANE)];
			uint32_t *keys_in = A.lkey, *keys_out = (uint32_t *) A.lq + 3 * A.prob_cap;
			int32_t *vals_in = A.lq;
			vals_out = A.lq + 2 * A.prob_cap;
			size_t tmp_bytes = 0;
			if(rocprim::radix_sort_pairs_desc((void *) nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, nl, 0u, 28u, stream) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs_desc (size query) failed"); return KMAHIP_EDEVICE; }
			// the lanes' stretches of move matrix: as many workgroups per class as this pass has work for
			size_t lane_e_bytes = 0;
			for(int j = 0; j < LT_LCLS; ++j) {
				lg[j].e_off = lane_e_bytes;
				lane_e_bytes += (size_t) std::min<unsigned long long>((unsigned long long) lg[j].wgs, (c[LCI(LC_LCNT + j)] + 63) / 64) * 64 * lg[j].estride;
			}
			if((rc = lt_reserve(ws, 9, std::max<size_t>(tmp_bytes, 16))) || (rc = lt_reserve(ws, 10, std::max<size_t>(lane_e_bytes, 16)))) return rc;
			if(rocprim::radix_sort_pairs_desc(ws->lt_buf[9], tmp_bytes, keys_in, keys_out, vals_in, vals_out, nl, 0u, 28u, stream) != hipSuccess) { kmahip_set_error("rocprim::radix_sort_pairs_desc failed"); return KMAHIP_EDEVICE; }
			stage("lane sort");
		}
		// (three side streams and the caller's: the runtime maps streams onto four hardware queues, and streams that share a queue
		// run their kernels one after the other -- with five side streams the stage took 1.5 to 3.6 s for the same million reads,
		// depending on which streams the process had made before)
		// (the third side stream is the seeding's: the sweeps use the caller's stream and two more)
		hipStream_t s1 = stream, s2 = stream, s3 = stream;
		hipEvent_t fork = nullptr, join[NSIDE] = {nullptr, nullptr, nullptr};
		if(!dbg) {
			HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
			HIP_TRY(hipEventRecord(fork, stream));
			for(int x = 0; x < NSIDE; ++x) HIP_TRY(hipStreamWaitEvent(side[x], fork, 0));
			s1 = side[0]; s2 = side[1]; s3 = side[1];
		}
		// the stream of a wave-per-problem class (KMAHIP_LT_XSTREAMS: a digit per class 4 .. 16, 0 = the caller's stream, 1 / 2 = side streams)
		auto xs = [&](int cls, hipStream_t dflt) -> hipStream_t {
			const char *map = getenv("KMAHIP_LT_XSTREAMS");
			if(!map || dbg || (int) strlen(map) <= cls - 4) return dflt;
			const int x = map[cls - 4] - '0';
			return x == 0 ? stream : x == 1 ? s1 : x == 2 ? s2 : dflt;
		};
		if(c[LCI(LC_CNT + 0)]) { hipLaunchKernelGGL((lt_dp_kernel<8>), wgs(0, 32, dp_wgs), dim3(256), 0, stream, A, 0); stage("dp<8>"); }
		if(c[LCI(LC_CNT + 1)]) { hipLaunchKernelGGL((lt_dp_kernel<16>), wgs(1, 16, dp_wgs), dim3(256), 0, stream, A, 1); stage("dp<16>"); }
		if(c[LCI(LC_CNT + 2)]) { hipLaunchKernelGGL((lt_dp_kernel<32>), wgs(2, 8, dp_wgs), dim3(256), 0, stream, A, 2); stage("dp<32>"); }
		if(c[LCI(LC_CNT + 3)]) { hipLaunchKernelGGL((lt_dp_kernel<64>), wgs(3, 4, dp_wgs), dim3(256), 0, stream, A, 3); stage("dp<64>"); }
		if(c[LCI(LC_CNT + 4)]) { hipLaunchKernelGGL((lt_dpx_kernel<2, false, false>), wgs(4, 1, 4 * dpx_wgs), dim3(64), LT_XE_LDS, xs(4, s1), A); stage("dpx<2, full>"); }
		if(c[LCI(LC_CNT + 5)]) { hipLaunchKernelGGL((lt_dpx_kernel<4, false, false>), wgs(5, 1, 4 * dpx_wgs), dim3(64), LT_XE_LDS, xs(5, s2), A); stage("dpx<4, full>"); }
		if(c[LCI(LC_CNT + 6)]) { hipLaunchKernelGGL((lt_dpx_kernel<2, true, false>), wgs(6, 1, 4 * dpx_wgs), dim3(64), LT_XE_LDS, xs(6, s1), A); stage("dpx<2, banded>"); }
		if(c[LCI(LC_CNT + 7)]) { hipLaunchKernelGGL((lt_dpx_kernel<4, true, false>), wgs(7, 1, 4 * dpx_wgs), dim3(64), LT_XE_LDS, xs(7, s2), A); stage("dpx<4, banded>"); }
		// (the HBM variants and the one-lane class share the per-workgroup scratch of dpx_wgs workgroups: one stream, in turn)
		if(c[LCI(LC_CNT + 8)]) { hipLaunchKernelGGL(lt_serial_kernel, wgs(8, 1, dpx_wgs), dim3(64), 0, s3, A); stage("serial"); }
		if(c[LCI(LC_CNT + 9)]) { hipLaunchKernelGGL((lt_dpx_kernel<2, false, true>), wgs(9, 1, dpx_wgs), dim3(64), 0, s3, A); stage("dpx<2, full, HBM>"); }
		if(c[LCI(LC_CNT + 10)]) { hipLaunchKernelGGL((lt_dpx_kernel<4, false, true>), wgs(10, 1, dpx_wgs), dim3(64), 0, s3, A); stage("dpx<4, full, HBM>"); }
		if(c[LCI(LC_CNT + 11)]) { hipLaunchKernelGGL((lt_dpx_kernel<2, true, true>), wgs(11, 1, dpx_wgs), dim3(64), 0, s3, A); stage("dpx<2, banded, HBM>"); }
		if(c[LCI(LC_CNT + 12)]) { hipLaunchKernelGGL((lt_dpx_kernel<4, true, true>), wgs(12, 1, dpx_wgs), dim3(64), 0, s3, A); stage("dpx<4, banded, HBM>"); }
		if(c[LCI(LC_CNT + 15)]) { hipLaunchKernelGGL((lt_dpx_kernel<8, true, true>), wgs(15, 1, dpx_wgs), dim3(64), 0, s3, A); stage("dpx<8, banded, HBM>"); }
		if(c[LCI(LC_CNT + 16)]) { hipLaunchKernelGGL((lt_dpx_kernel<16, true, true>), wgs(16, 1, dpx_wgs), dim3(64), 0, s3, A); stage("dpx<16, banded, HBM>"); }
		if(c[LCI(LC_LANE)]) {
			size_t off = 0;
			for(int j = LT_LCLS - 1; j >= 0; --j) {
				const unsigned long long cnt = c[LCI(LC_LCNT + j)];
				if(!cnt) continue;
				LaneArgs La;
				La.queue = vals_out + off; La.count = cnt; La.R = lg[j].g.R; La.RQ = lg[j].g.RQ; La.TW = lg[j].g.TW; La.ecap = lg[j].g.ecap; La.estride = (int) lg[j].estride;
				La.E = (uint8_t *) ws->lt_buf[10] + lg[j].e_off;
				La.ablate = getenv("KMAHIP_LT_ABLATE") ? atoi(getenv("KMAHIP_LT_ABLATE")) : 0;
				const unsigned grid = (unsigned) std::min<unsigned long long>((unsigned long long) lg[j].wgs, (cnt + 63) / 64);
				hipStream_t ls = j <= 4 ? stream : (j == LT_LCLS - 1 ? s2 : s1);          // (tools/c4_time.py 200 000: class 4 on the second side stream 230 ms, here 210)
				if(const char *map = getenv("KMAHIP_LT_STREAMS")) {       // diagnosis: a digit per lane class, 0 = the caller's stream, 1-3 = side streams
					if((int) strlen(map) > j && !dbg) { const int x = map[j] - '0'; const hipStream_t all[4] = {stream, s1, s2, s3}; if(x >= 0 && x < 4) ls = all[x]; }
				}
				const void *fn = j < LT_LFULL ? (simple_sc ? (const void *) lt_lane_kernel<true> : (const void *) lt_lane_kernel<false>)
				                       : (simple_sc ? (const void *) lt_lane_band_kernel<true> : (const void *) lt_lane_band_kernel<false>);
				if(lg[j].lds > 65536) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lg[j].lds));
				if((j <= 3 || (j == 4 && reg_wide)) && reg_rows) {
					// rows of up to 64 cells: the row in registers (lt_reg_kernel); no LDS but the staged template rows and query codes
					const size_t lds = (size_t) (32 + lg[j].g.TW * 64) * 4 + (size_t) lg[j].g.RQ * 64;
					const unsigned rgrid = grid;          // (the move matrices' scratch is sized for lg[j].wgs workgroups)
					if(j == 0) hipLaunchKernelGGL((lt_reg_kernel<16>), dim3(rgrid), dim3(64), lds, ls, A, La);
					else if(j == 1) hipLaunchKernelGGL((lt_reg_kernel<32>), dim3(rgrid), dim3(64), lds, ls, A, La);
					else if(j == 2) hipLaunchKernelGGL((lt_reg_kernel<48>), dim3(rgrid), dim3(64), lds, ls, A, La);
					else if(j == 3) hipLaunchKernelGGL((lt_reg_kernel<64>), dim3(rgrid), dim3(64), lds, ls, A, La);
					else hipLaunchKernelGGL((lt_reg_kernel<128>), dim3(rgrid), dim3(64), lds, ls, A, La);
				}
				else if((j == LT_LFULL || j == LT_LFULL + 1) && reg_band) {
					const size_t lds = (size_t) (32 + lg[j].g.TW * 64) * 4 + (size_t) lg[j].g.RQ * 32;
					if(j == LT_LFULL) hipLaunchKernelGGL((lt_regband_kernel<72>), dim3(grid), dim3(64), lds, ls, A, La);
					else hipLaunchKernelGGL((lt_regband_kernel<96>), dim3(grid), dim3(64), lds, ls, A, La);
				}
				else if(j < LT_LFULL) { if(simple_sc) hipLaunchKernelGGL(lt_lane_kernel<true>, dim3(grid), dim3(64), lg[j].lds, ls, A, La); else hipLaunchKernelGGL(lt_lane_kernel<false>, dim3(grid), dim3(64), lg[j].lds, ls, A, La); }
				else { if(simple_sc) hipLaunchKernelGGL(lt_lane_band_kernel<true>, dim3(grid), dim3(64), lg[j].lds, ls, A, La); else hipLaunchKernelGGL(lt_lane_band_kernel<false>, dim3(grid), dim3(64), lg[j].lds, ls, A, La); }
				if(dbg) {
					std::vector<uint32_t> hk((size_t) cnt);
					(void) hipMemcpy(hk.data(), (uint32_t *) A.lq + 3 * A.prob_cap + off, (size_t) cnt * 4, hipMemcpyDeviceToHost);
					unsigned long long turns = 0, rounds = 0;
					for(size_t x = 0; x < hk.size(); ++x) { turns += hk[x] & 0xFFFFFFu; if(x % 64 == 0) rounds += hk[x] & 0xFFFFFFu; }
					fprintf(stderr, "[kmahip] longtrace: lane class %d: %llu problems, %llu turns of their sweeps, %llu turns of the wavefronts (x 64 = %.2f of them used), longest %u\n",
					        j, cnt, turns, rounds, rounds ? (double) turns / (64.0 * rounds) : 0.0, hk.empty() ? 0u : hk[0] & 0xFFFFFFu);
				}
				stage("lane class");
				off += (size_t) cnt;
			}
		}
		if(!dbg) {
			for(int x = 0; x < NSIDE; ++x) {
				HIP_TRY(hipEventCreateWithFlags(&join[x], hipEventDisableTiming));
				HIP_TRY(hipEventRecord(join[x], side[x]));
				HIP_TRY(hipStreamWaitEvent(stream, join[x], 0));
			}
		}
		hipLaunchKernelGGL(lt_finish_kernel, dim3((unsigned) std::min<int64_t>(fin_wgs, nb)), dim3(64), 0, stream, A);
		stage("finish");
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(hc + (size_t) (2 + px) * LCI(LC_N), A.counters, (size_t) LCI(LC_N) * 8, hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipEventRecord(ev.fin[px], stream));
		fin_pending[px] = true; fin_nb[px] = nb;
		if(fork) { (void) hipEventDestroy(fork); for(int x = 0; x < NSIDE; ++x) (void) hipEventDestroy(join[x]); }
		r0 += nb;
		if(piped && r0 < n) {
			// the next pass's seeding beside this pass's sweeps, in the other set -- once that set's last pass is through
			if((rc = fin_check(1 - px)) || (rc = set_args(1 - px, r0, std::min<int64_t>(B, n - r0))) || (rc = launch_seed(1 - px, side[2]))) return rc;
			seeded = true;
			px = 1 - px;
		} else if((rc = fin_check(px))) return rc;
	}
	if((rc = fin_check(0)) || (rc = fin_check(1))) return rc;
	HIP_TRY(hipStreamSynchronize(stream));
	// the caller reads the pool top through the workspace counters like after trace_kernel: [0] = runs used, [1] = status
	if(score_mode) return KMAHIP_OK;          // (no runs; the workspace's status word is stage 3a's)
	unsigned long long fin[2] = {0, 0};
	HIP_TRY(hipMemcpy(&fin[0], counters + LCI(LC_OUT), 8, hipMemcpyDeviceToHost));
	if((int64_t) fin[0] > out->ops_cap) fin[1] = 2;
	if(!ws->counters) { HIP_TRY(hipMalloc((void **) &ws->counters, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); HIP_TRY(hipMemset(ws->counters, 0, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); }
	HIP_TRY(hipMemcpy(ws->counters, fin, sizeof fin, hipMemcpyHostToDevice));
	return KMAHIP_OK;
}
