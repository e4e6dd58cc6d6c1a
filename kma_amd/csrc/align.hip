// align.hip -- stage 3a of KMA on gfx950: for every (read, candidate template)
// task find the maximal exact matches against the template position index,
// chain them, stitch the chain with global / banded Needleman-Wunsch on the gaps
// and tails, then filter and pick per read.  Behaviour restated from
//   KMA_score        align.c:509-748      chainSeeds      chain.c:79-260
//   lead/trailTailAln align.c:53-212      NW_score        nw.c:642-890
//   alnFragsSE       alnfrags.c:1052-1218 NW_band_score   nw.c:892-1188
//   update_Scores    updatescores.c:203-298
//
// Structure (new): seed_tasks_kernel finds the MEMs of every
// task at 8 waves / SIMD (one lane per task, up to 4 MEMs handed over); the main
// kernel takes tasks 64 at a time from a device-wide counter, one lane per task:
// chaining and stitching per lane (MEM arrays in an HBM scratch laid out
// [index][lane]), every DP problem wider than one column deferred into per-wave LDS
// queues and solved by the whole wave on anti-diagonal sweeps (nw_coop<8|16|64>: up to
// 63 query columns, nw_coop_x: up to 255 columns, full and banded), the results added
// back per lane.  The trace kernel of stage 3c (KMA(), align.c:214-507) is further down.
// The DP keeps ONE row in place and no traceback
// matrix: the reference fills a byte matrix E and then walks it only to count
// (len, match, tGaps, qGaps); because that walk from a cell visits only cells
// filled earlier, the counts of the walk from every cell are carried beside
// D/P/Q as three packed 21-bit counters ("count-carrying DP"), including the
// walk's quirk that a gap run ends on the first cell with EITHER may-open bit.
#include "kmahip_internal.h"
#include "dna_dev.h"
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <climits>

#ifdef KMAHIP_DIAG
__device__ unsigned long long *g_diag_hist = nullptr;
#endif

namespace {

constexpr int ATHREADS = 256;
constexpr uint64_t MA = 1ull;             // one diagonal step
constexpr uint64_t TG = 1ull << 21;       // one gap-in-template step (query consumed)
constexpr uint64_t QG = 1ull << 42;       // one gap-in-query step (template consumed)

struct AlignArgs {
	DevDB db;
	int64_t n_reads;
	const uint64_t *seq;
	const int64_t *seq_off;
	const int32_t *len;
	const int32_t *N;
	const int64_t *N_off;
	const int32_t *rc_flag;
	const int32_t *flag;
	const int64_t *T_off;
	const int32_t *T;
	const int32_t *q_start, *q_end;   // query bounds per read (default-mode records), NULL: whole reads
	int M, MM, U, W1;
	int d[25];
	int minlen, mq;
	double scoreT, mrc;
	// per task
	int32_t *t_score, *t_alen, *t_start, *t_end, *t_tmpl;
	double *t_norm;
	const int32_t *t_rec;    // record owning each task (expanded from T_off by task_map_kernel)
	int32_t *seed_n;         // per seed slot, written by seed_tasks_kernel: number of MEMs found (0..SEEDS), -1: seed in the main kernel
	uint2 *seed_mem;         // SEEDS per slot: x = tS (1-based), y = qS | (length << 16)
	int seed_slots;          // seed slots per task: 1 (single end) or 2 (paired records: both mates of a couple)
	int *xq;                 // spill queues for deferred DP problems, per wavefront 4 classes x xq_cap entries of QENT ints (long reads)
	int xq_cap;
	int64_t tasks_cap;       // capacity of T / the task scratch / the caller's hit arrays: a candidate list that outgrew it (stage 2 set
	                         // status 2 and still wrote the full offsets) is not touched, every kernel returns at once
	// scratch
	int32_t *s32;
	uint64_t *s64;
	int64_t lanes;
	int mem_cap;     // MEM slots per lane (arrays hold mem_cap + 1)
	int ncols;       // DP columns per lane
	// paired-end mode: the 'reads' of the candidate CSR are stage-2 records (2 per pair)
	const int32_t *rec_mate, *rec_rc;
	int pe_mode, Wl, PE;
	unsigned long long *counters;   // [1] status
	int stats;
	int ablate;      // diagnostic builds only (KMAHIP_DIAG): 1 skip DP, 2 skip seeding, 4 skip chaining
	int gap_m_max;   // Lane::gap_m_max
	int long_min;    // > 0: the tasks of reads this long (strand known, no N's) are left to the long-read pipeline (long_routed)
	int64_t *slow_list;      // tasks align_fast_kernel hands on (counters[2] of them); the general kernel run over a list takes its tasks from it
	int use_list;            // align_tasks_kernel: the tasks are slow_list[0 .. counters[2]), handed out through counters[8]
	int *dq; int64_t dq_cap; // align_fast_kernel's DP problems: four class queues of dq_cap entries of QENT ints (counters[AC_DQ + class])
	int32_t *pend;           // its tasks that wait for queued problems (counters[AC_PEND]) ...
	int *part;               // ... and their sums so far, PART_INTS per TASK
	int per_round;           // tasks a wavefront takes per round (64; fewer over the list, whose tasks all bring DP problems for the wave's queues)
};

constexpr int SEEDS = 4;        // MEMs per task the seeding kernel hands over (a 150 bp read has 1-3)

struct Lane {
	int32_t *s32;
	uint64_t *s64;
	int32_t *r32;              // this lane's two int DP rows (2 * ncols) ...
	uint64_t *r64;             // ... and its two counter rows
	int64_t lanes;
	int cap1;    // mem_cap + 1
	int ncols;
	const int *d;   // 25 ints in LDS
	int M, MM, U, W1;
	uint32_t *wide;            // this wave's LDS row slots (WSLOTS x 4 planes x WCOLS words)
	int *queue;                // this wave's deferred wide-DP queue in LDS: [0] = count, entries of QENT ints from [1]
	int *xq;                   // this wave's spill queues in HBM (long reads: hundreds of link problems per task), 4 classes x xq_cap entries
	int *xq_cnt;               // their fill counts (LDS, 4 ints)
	int xq_cap;
	int q_at, q_mate;          // context of the kma_score call being run (template id, mate slot) for queue entries
	int64_t q_rd;              // read index of the query being aligned
	int ablate;
	int gap_m_max;             // >= 0: DP problems with a provably ungapped answer skip the matrix (nw_diagonal), up to this many mismatches between seeds
	int diag_uniform;          // d[0][0] == d[1][1] == d[2][2] == d[3][3]: a MEM (never holds an N) scores span * d[0][0]
	unsigned long long *cnt;   // work counters (stats launches only): [3] lookups [4] MEM bases [5] DP cells [6] tasks
};

// MEM arrays: 0 tS, 1 tE, 2 qS, 3 qE, 4 weight, 5 score, 6 next
#define MEMA(L, a, m) (L).s32[((int64_t) ((a) * (L).cap1 + (m))) * (L).lanes]
// DP rows of the problems too wide for the cooperative queues: contiguous per lane (a lane that gets one walks it alone,
// so neighbouring columns should share cache lines; strided by the lane count every cell was its own line and a single
// 150 x 130 tail held its kernel for 5 ms)
#define ROWD(L, n) (L).r32[(n)]
#define ROWP(L, n) (L).r32[(L).ncols + (n)]
#define ROWTD(L, n) (L).r64[(n)]
#define ROWTP(L, n) (L).r64[(L).ncols + (n)]

constexpr int QCAP = 8;             // deferred wide problems (17..63 columns) per wave and task round
constexpr int QCAPN = 20;           // deferred narrow problems (9..16 columns)
constexpr int QCAPT = 32;           // deferred tiny problems (2..8 columns): eight side by side per cooperative call
constexpr int QENT = 12;            // ints per queue entry
constexpr int QCAPX = 12;           // deferred extra-wide problems (64..255 columns): one per cooperative call, XC columns per lane
constexpr int XC = 4;
constexpr int QINTS = 4 + (QCAP + QCAPN + QCAPT + QCAPX) * QENT;   // ints of one wave's four queues
constexpr int TBUF = 1024;          // staged template bases per cooperative problem
constexpr int WCOLS = 64;           // LDS row slots for "wide" problems: 17..63 query columns
constexpr int WSLOTS = 2;           // slots per wave

struct GRows {
	typedef uint64_t ST;
	static constexpr int SH = 21;
	int32_t *d, *p;
	uint64_t *td, *tp;
	int64_t stride;
	__device__ __forceinline__ int32_t &D(int n) const { return d[(int64_t) n * stride]; }
	__device__ __forceinline__ int32_t &P(int n) const { return p[(int64_t) n * stride]; }
	__device__ __forceinline__ uint64_t &TD(int n) const { return td[(int64_t) n * stride]; }
	__device__ __forceinline__ uint64_t &TP(int n) const { return tp[(int64_t) n * stride]; }
};

struct LRows {
	typedef uint32_t ST;
	static constexpr int SH = 10;
	uint32_t *base;   // planes D, P, TD, TP of WCOLS words each, one lane at a time
	__device__ __forceinline__ int32_t &D(int n) const { return ((int32_t *) base)[n]; }
	__device__ __forceinline__ int32_t &P(int n) const { return ((int32_t *) base)[WCOLS + n]; }
	__device__ __forceinline__ uint32_t &TD(int n) const { return base[2 * WCOLS + n]; }
	__device__ __forceinline__ uint32_t &TP(int n) const { return base[3 * WCOLS + n]; }
};

template <int SH>
__device__ __forceinline__ Aln aln_from(int score, uint64_t st) {
	Aln a;
	const uint64_t fm = (1ull << SH) - 1;
	a.score = score; a.pos = 0;
	a.match = (int) (st & fm); a.tGaps = (int) ((st >> SH) & fm); a.qGaps = (int) ((st >> (2 * SH)) & fm);
	a.len = a.match + a.tGaps + a.qGaps;
	return a;
}

__device__ Aln nw_degenerate(int t_len, int q_len, int U, int W1) {
	Aln s = {0, 0, 0, 0, 0, 0};
	if(t_len == q_len) return s;
	if(t_len == 0) { s.len = q_len; s.tGaps = q_len; s.score = W1 + (q_len - 1) * U; }
	else { s.len = t_len; s.qGaps = t_len; s.score = W1 + (t_len - 1) * U; }
	return s;
}

// NW_score, nw.c:642-890 (mode k: 0 global, -1/-2 free leading template / both, +1/+2 free trailing)
// R = where the single DP row lives (GRows: HBM scratch, 21-bit walk counters).
template <class R>
__device__ Aln nw_full(const R r, const Lane &L, const uint64_t *ts, int tlen_total, const QView &q, int k,
                       int t_s, int t_e, int q_s, int q_e) {
	typedef typename R::ST ST;
	const ST MA = 1, TG = (ST) 1 << R::SH, QG = (ST) 1 << (2 * R::SH);
	const int U = L.U, W1 = L.W1;
	int t_len = t_e - t_s;
	const int q_len = q_e - q_s;
	if(t_len < 0) t_len += tlen_total;
	if(t_len == 0 || q_len == 0) return nw_degenerate(t_len, q_len, U, W1);
	const int low = (t_len + q_len) * (L.MM + U + W1);
	if(L.cnt) atomicAdd(&L.cnt[5], (unsigned long long) t_len * q_len);
	if(k == 2) {
		for(int n = 0; n <= q_len; ++n) { r.D(n) = 0; r.P(n) = low; r.TD(n) = 0; r.TP(n) = 0; }
	} else {
		for(int n = 0; n < q_len; ++n) {
			r.D(n) = W1 + (q_len - 1 - n) * U; r.P(n) = low;
			r.TD(n) = TG * (ST) (q_len - n); r.TP(n) = 0;
		}
		r.D(q_len) = 0; r.P(q_len) = 0; r.TD(q_len) = 0; r.TP(q_len) = 0;
	}
	int best = low;
	ST bestTD = 0;
	QCursor qc; qc.blk = -1; qc.w = 0;
	int npos = t_e - 1;
	for(int m = t_len - 1; m >= 0; --m, --npos) {
		if(npos < 0) npos = tlen_total - 1;
		int diagD = r.D(q_len);
		ST diagTD = r.TD(q_len);
		int Dright = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
		ST TDright = (0 < k) ? (ST) 0 : QG * (ST) (t_len - m);
		ST TQright = 0;
		r.D(q_len) = Dright; r.TD(q_len) = TDright;
		int Qprev = low;
		const int *drow = L.d + 5 * tn(ts, npos);
		for(int n = q_len - 1; n >= 0; --n) {
			const int Dp = r.D(n), Pp = r.P(n);
			const ST TDp = r.TD(n), TPp = r.TP(n);
			int Q = Dright + W1, P = Dp + W1, D, mv;
			bool ob = false;
			if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
			int x = Qprev + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else ob = true;
			x = Pp + U;
			if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else ob = true;
			x = diagD + drow[qc.code(q, q_s, n)];
			if(D <= x) { D = x; mv = 1; }
			const ST TQ = TG + (ob ? TDright : TQright);
			const ST TP = QG + (ob ? TDp : TPp);
			const ST TD = (mv == 1) ? (MA + diagTD) : (mv >= 4 ? TP : TQ);
			r.D(n) = D; r.P(n) = P; r.TD(n) = TD; r.TP(n) = TP;
			diagD = Dp; diagTD = TDp; Dright = D; TDright = TD; TQright = TQ; Qprev = Q;
		}
		if(k < 0 && best < Dright) { best = Dright; bestTD = TDright; }
	}
	if(k < 0) {
		if(k == -2) {
			for(int n = 0; n < q_len; ++n) {
				const int D = r.D(n);
				if(best <= D) { best = D; bestTD = r.TD(n); }
			}
		}
		return aln_from<R::SH>(best, (uint64_t) bestTD);
	}
	return aln_from<R::SH>(r.D(0), (uint64_t) r.TD(0));
}

// NW_score with a single query column (the 1 x 1 gap between two MEMs split by a mismatch is by far the most
// common DP problem): same recurrences as nw_full, all state in scalars.
__device__ Aln nw_col1(const Lane &L, const uint64_t *ts, int tlen_total, const QView &q, int k, int t_s, int t_e, int q_s) {
	const int U = L.U, W1 = L.W1;
	int t_len = t_e - t_s;
	if(t_len < 0) t_len += tlen_total;
	const uint32_t MA = 1u, TG = 1u << 10, QG = 1u << 20;
	const int low = (t_len + 1) * (L.MM + U + W1);
	if(L.cnt) atomicAdd(&L.cnt[5], (unsigned long long) t_len);
	const int qc = qn(q, q_s);
	// boundary row m = t_len: column 0 and the boundary column 1
	int D = (k == 2) ? 0 : W1, P = low, Db = 0;
	uint32_t TD = (k == 2) ? 0u : TG, TP = 0, TDb = 0;
	int best = low;
	uint32_t bestTD = 0;
	int npos = t_e - 1;
	for(int m = t_len - 1; m >= 0; --m, --npos) {
		if(npos < 0) npos = tlen_total - 1;
		const int *drow = L.d + 5 * tn(ts, npos);
		const int Dright = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
		const uint32_t TDright = (0 < k) ? 0u : QG * (uint32_t) (t_len - m);
		int Q = Dright + W1, Pn = D + W1, Dn, mv;
		bool ob = false;
		if(Q < Pn) { Dn = Pn; mv = 4; } else { Dn = Q; mv = 2; }
		int x = low + U;
		if(Q < x) { Q = x; if(Dn <= x) { Dn = x; mv = 3; } } else ob = true;
		x = P + U;
		if(Pn < x) { Pn = x; if(Dn <= x) { Dn = x; mv = 5; } } else ob = true;
		x = Db + drow[qc];
		if(Dn <= x) { Dn = x; mv = 1; }
		const uint32_t TQ = TG + (ob ? TDright : 0u);
		const uint32_t TPn = QG + (ob ? TD : TP);
		const uint32_t TDn = (mv == 1) ? (MA + TDb) : (mv >= 4 ? TPn : TQ);
		Db = Dright; TDb = TDright;
		D = Dn; P = Pn; TD = TDn; TP = TPn;
		if(k < 0 && best < D) { best = D; bestTD = TD; }
	}
	if(k < 0) {
		if(k == -2 && best <= D) { best = D; bestTD = TD; }
		return aln_from<10>(best, (uint64_t) bestTD);
	}
	return aln_from<10>(D, (uint64_t) TD);
}

// NW_band_score, nw.c:892-1188. Column n of row m is column n-1 of row m+1.
__device__ Aln nw_band(const Lane &L, const uint64_t *ts, int tlen_total, const QView &q, int k,
                       int t_s, int t_e, int q_s, int q_e, int band) {
	const int U = L.U, W1 = L.W1;
	int t_len = t_e - t_s;
	const int q_len = q_e - q_s;
	if(t_len < 0) t_len += tlen_total;
	if(t_len == 0 || q_len == 0) return nw_degenerate(t_len, q_len, U, W1);
	if(band & 1) ++band;
	const int half = band >> 1, bq = band + 1;
	const int low = (t_len + q_len) * (L.MM + U + W1);
	if(L.cnt) atomicAdd(&L.cnt[5], (unsigned long long) t_len * bq);
	int c = (t_len + q_len) >> 1;
	int sn = q_len - 1 - (c - half);
	// rows are never read outside what the previous row wrote, except for cells the
	// reference itself leaves to whatever the buffers held; start from zeroes like the oracle
	for(int n = 0; n <= bq + 1; ++n) { ROWD(L, n) = 0; ROWP(L, n) = 0; ROWTD(L, n) = 0; ROWTP(L, n) = 0; }
	if(k != 2) {
		for(int n = sn - 1; n >= 0; --n) { ROWD(L, n) = W1 + (sn - n - 1) * U; ROWP(L, n) = low; ROWTD(L, n) = TG * (uint64_t) (sn - n); }
		ROWD(L, sn) = 0; ROWP(L, sn) = 0; ROWTD(L, sn) = 0;
	} else {
		for(int n = sn; n >= 0; --n) { ROWD(L, n) = 0; ROWP(L, n) = low; ROWTD(L, n) = 0; }
	}
	int best = low, bm = 0, en = 0, n = 0;
	QCursor qc; qc.blk = -1; qc.w = 0;
	uint64_t bestTD = 0;
	int npos = t_e - 1;
	for(int m = t_len - 1; m >= 0; --m, --npos, --c) {
		if(npos < 0) npos = tlen_total - 1;
		int sq = c + half, eq = c - half;
		if(eq < 0) { eq = 0; ++en; } else en = 0;
		int Qprev = low;
		int Dright;
		uint64_t TDright, TQright = 0;
		if(sq < q_len - 1) {
			sn = bq - 1;
			// cell bq: D = low, E = 37 (query-gap extension carrying its open bit)
			Dright = low;
			TDright = QG + ROWTD(L, bq - 1);
			ROWD(L, bq) = Dright; ROWTD(L, bq) = TDright;
		} else {
			sq = q_len - 1; sn = en + (q_len - eq);
			Dright = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
			TDright = (0 < k) ? 0ull : (QG + ROWTD(L, sn - 1));
			ROWD(L, sn) = Dright; ROWTD(L, sn) = TDright;
			--sn;
		}
		const int *drow = L.d + 5 * tn(ts, npos);
		int qp = sq;
		for(n = sn; n > en; --qp, --n) {
			const int Dp1 = ROWD(L, n - 1), Pp1 = ROWP(L, n - 1), Dp = ROWD(L, n);
			const uint64_t TDp1 = ROWTD(L, n - 1), TPp1 = ROWTP(L, n - 1), TDp = ROWTD(L, n);
			int Q = Dright + W1, P = Dp1 + W1, D, mv;
			bool ob = false;
			if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
			int x = Qprev + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else ob = true;
			x = Pp1 + U;
			if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else ob = true;
			x = Dp + drow[qc.code(q, q_s, qp)];
			if(D <= x) { D = x; mv = 1; }
			const uint64_t TQ = TG + (ob ? TDright : TQright);
			const uint64_t TP = QG + (ob ? TDp1 : TPp1);
			const uint64_t TD = (mv == 1) ? (MA + TDp) : (mv >= 4 ? TP : TQ);
			ROWD(L, n) = D; ROWP(L, n) = P; ROWTD(L, n) = TD; ROWTP(L, n) = TP;
			Dright = D; TDright = TD; TQright = TQ; Qprev = Q;
		}
		{	// band edge (nw.c:1079-1105): no gap-in-query state
			const int Dp = ROWD(L, n);
			const uint64_t TDp = ROWTD(L, n);
			int Q = Dright + W1, mv;
			bool ob = false;
			const int x = Qprev + U;
			if(Q < x) { Q = x; mv = 3; } else { mv = 2; ob = true; }
			int D = Dp + drow[qc.code(q, q_s, qp)];
			if(Q <= D) mv = 1; else D = Q;
			const uint64_t TQ = TG + (ob ? TDright : TQright);
			const uint64_t TD = (mv == 1) ? (MA + TDp) : TQ;
			ROWD(L, n) = D; ROWP(L, n) = low; ROWTD(L, n) = TD; ROWTP(L, n) = 0;
			if(eq == 0 && k < 0 && best < D) { best = D; bestTD = TD; bm = m; }
		}
	}
	if(bm == 0) { best = ROWD(L, en); bestTD = ROWTD(L, en); }
	if(k == -2) {
		for(n = en; n < bq; ++n) {
			const int D = ROWD(L, n);
			if(best <= D) { best = D; bestTD = ROWTD(L, n); }
		}
	}
	return aln_from<21>(best, bestTD);
}

// hand a DP problem to the wave: class 0 wide (17..63 columns), 1 narrow (9..16), 2 tiny (2..8), 3 extra-wide / banded.
// First the LDS queue of the class, then (long reads) its spill queue in HBM; false: the caller solves it in its lane.
__device__ __forceinline__ bool dp_enqueue(const Lane &L, int cls, int k, int t_s, int t_e, int q_s, int q_e, int rc, int e11) {
	int *qu = cls == 2 ? L.queue + (2 + (QCAP + QCAPN) * QENT) : cls == 1 ? L.queue + (1 + QCAP * QENT) : cls == 0 ? L.queue : L.queue + (3 + (QCAP + QCAPN + QCAPT) * QENT);
	const int cap = cls == 2 ? QCAPT : cls == 1 ? QCAPN : cls == 0 ? QCAP : QCAPX;
	int *e = nullptr;
	const int slot = atomicAdd(&qu[0], 1);
#ifdef KMAHIP_DIAG
	if(L.cnt && g_diag_hist && cls < 3) atomicAdd(&g_diag_hist[200 + (cls == 2 ? 0 : cls == 1 ? 1 : 2) + (slot < cap ? 0 : 4)], 1ull);
#endif
	if(slot < cap) e = qu + 1 + slot * QENT;
	else if(L.xq_cap > 0) {
		const int s2 = atomicAdd(&L.xq_cnt[cls], 1);
		if(s2 < L.xq_cap) e = L.xq + ((size_t) cls * L.xq_cap + s2) * QENT;
	}
	if(!e) return false;
	e[0] = (int) (threadIdx.x & 63); e[1] = L.q_mate; e[2] = k; e[3] = t_s; e[4] = t_e; e[5] = q_s; e[6] = q_e;
	e[7] = L.q_at; e[8] = (int) (L.q_rd & 0xFFFFFFFFll); e[9] = (int) (L.q_rd >> 32); e[10] = rc; e[11] = e11;
	return true;
}

// mismatches between g read bases from qp and g template bases from tp (neither side holds an N; both word arrays are padded)
__device__ __forceinline__ int diag_mism(const uint64_t *ts, const QView &q, int tp, int qp, int g) {
	int m = 0;
	for(int o = 0; o < g; o += 32) {
		const int step = min(32, g - o);
		uint64_t x = (qwin(q, qp + o) ^ win2(ts, tp + o)) >> (64 - 2 * step);
		x = (x | (x >> 1)) & 0x5555555555555555ull;
		m += __popcll((long long) x);
	}
	return m;
}

// A DP problem whose only optimal alignment is the diagonal (see the proof at diag_emit below): score and counts without the
// matrix. gap_m_max < 0: switched off (the score matrix is not plain match / mismatch).
__device__ __forceinline__ bool nw_diagonal(const Lane &L, const uint64_t *ts, const QView &q, int k, int t_s, int t_e, int q_s, int q_e,
                                            int tspan, Aln &out) {
	const int g = q_e - q_s;
	if(L.gap_m_max < 0 || q.nN || t_e - t_s != tspan || g <= 0) return false;
	// the diagonal in question, the one next to it (tails: a gap of one in front of the seed side) and the mismatches allowed
	int tp, tp2, lim;
	if(k == 0) { if(tspan != g) return false; tp = t_s; tp2 = t_s; lim = L.gap_m_max; }
	else if(k == -1) { if(tspan <= g) return false; tp = t_e - g; tp2 = t_e - 1 - g; lim = 1; }
	else if(k == 1) { if(tspan <= g) return false; tp = t_s; tp2 = t_s + 1; lim = 1; }
	else return false;
	const int m = diag_mism(ts, q, tp, q_s, g);
	if(m > lim) return false;
	if(k != 0 && m == 1 && diag_mism(ts, q, tp2, q_s, g) == 0) return false;
	out.score = (g - m) * L.M + m * L.MM; out.len = g; out.match = g; out.tGaps = 0; out.qGaps = 0; out.pos = 0;
	return true;
}

__device__ __forceinline__ Aln nw_auto(const Lane &L, const uint64_t *ts, int t_len, const QView &q, int k,
                                       int t_s, int t_e, int q_s, int q_e, int tspan, int band) {
#ifdef KMAHIP_DIAG
	if(L.ablate & 1) return nw_degenerate(tspan, 0, L.U, L.W1);
	if((L.ablate & 8) && q_e - q_s > 16) return nw_degenerate(tspan, 0, L.U, L.W1);
	if((L.ablate & 16) && q_e - q_s == 1) return nw_degenerate(tspan, 0, L.U, L.W1);
	if((L.ablate & 32) && q_e - q_s > 1 && q_e - q_s <= 16) return nw_degenerate(tspan, 0, L.U, L.W1);
	if(L.cnt && g_diag_hist) {
		const int ql = q_e - q_s, b = min(63, ql);
		atomicAdd(&g_diag_hist[b], 1ull);
		atomicAdd(&g_diag_hist[64 + b], (unsigned long long) max(0, tspan) * ql);
		atomicAdd(&g_diag_hist[128 + (k + 2)], 1ull);
		atomicAdd(&g_diag_hist[136 + min(63, max(0, tspan) >> 2)], 1ull);
	}
#endif
	if(q_e - q_s <= band || tspan <= band) {
		const int ql = q_e - q_s;
		if(ql == 0 || tspan == 0) return nw_degenerate(tspan, ql, L.U, L.W1);      // nw.c:662-684
		{ Aln dg; if(nw_diagonal(L, ts, q, k, t_s, t_e, q_s, q_e, tspan, dg)) return dg; }
		if(ql == 1 && tspan < 998) return nw_col1(L, ts, t_len, q, k, t_s, t_e, q_s);
		if(L.queue && ql >= WCOLS && ql < 64 * XC && tspan <= TBUF && tspan + ql < 1000) {
			// long unaligned ends (a read that matches a template only in part): a lane walking those cells alone takes
			// milliseconds and holds its whole wave
			if(dp_enqueue(L, 3, k, t_s, t_e, q_s, q_e, q.rc, (k < 0) ? 1 : 0)) { Aln z = {0, 0, 0, 0, 0, 0}; return z; }
		}
		if(L.queue && ql < WCOLS && tspan + ql < 1000) {
			// The other small problems are handed to the whole wave (nw_coop, run after every lane has finished its
			// own work): a lane walking a DP alone keeps 63 lanes idle at ~10^2 cycles per cell. Results are only
			// ever summed into the alignment statistics, so the caller goes on with zeroes.
			const bool tiny = ql <= 8 && tspan < TBUF / 8;
			const bool narrow = !tiny && ql <= 16 && tspan < TBUF / 4;
			if(dp_enqueue(L, tiny ? 2 : narrow ? 1 : 0, k, t_s, t_e, q_s, q_e, q.rc, (k < 0) ? 1 : 0)) { Aln z = {0, 0, 0, 0, 0, 0}; return z; }
		}
#ifdef KMAHIP_DIAG
		if(L.cnt && g_diag_hist) { atomicAdd(&g_diag_hist[208], 1ull); atomicAdd(&g_diag_hist[209], (unsigned long long) max(0, tspan) * (q_e - q_s)); if(q_e - q_s >= WCOLS) atomicAdd(&g_diag_hist[210], 1ull); }
		if(L.ablate & 128) return nw_degenerate(tspan, 0, L.U, L.W1);     // ablation: nothing that misses the queues
#endif
		if(q_e - q_s < WCOLS && tspan + (q_e - q_s) < 1000) {
			// queue full (or deferral off): the row goes to one of the wave's LDS slots; the lanes of this wave that
			// are here together take turns, WSLOTS at a time (they run in lock step anyway)
			const unsigned long long here = __ballot(1);
			const int rank = __popcll(here & ((1ull << (threadIdx.x & 63)) - 1ull));
			const int total = __popcll(here);
			Aln res = {0, 0, 0, 0, 0, 0};
			for(int round = 0; round * WSLOTS < total; ++round) {
				if(rank / WSLOTS == round) {
					LRows lr;
					lr.base = L.wide + (rank % WSLOTS) * 4 * WCOLS;
					res = nw_full(lr, L, ts, t_len, q, k, t_s, t_e, q_s, q_e);
				}
			}
			return res;
		}
		GRows gr;
		gr.d = &ROWD(L, 0); gr.p = &ROWP(L, 0); gr.td = &ROWTD(L, 0); gr.tp = &ROWTP(L, 0); gr.stride = 1;
		return nw_full(gr, L, ts, t_len, q, k, t_s, t_e, q_s, q_e);
	}
	{
		// banded problems of the same size go to the same queue (a read end that matches nothing: both sides beyond the band),
		// unless the reference's last-row scan (k == -2) would read cells of its row buffer beyond the boundary column
		const int ql = q_e - q_s;
		int b2 = band + (band & 1);
		const int cfin = ((tspan + ql) >> 1) - (tspan - 1);
		const bool stale_scan = k == -2 && !(cfin + (b2 >> 1) < ql - 1);
		if(L.queue && ql < 64 * XC && tspan <= TBUF && tspan + ql < 1000 && band < (1 << 20) && !stale_scan) {
			if(dp_enqueue(L, 3, k, t_s, t_e, q_s, q_e, q.rc, ((k < 0) ? 1 : 0) | (band << 1))) { Aln z = {0, 0, 0, 0, 0, 0}; return z; }
		}
	}
	return nw_band(L, ts, t_len, q, k, t_s, t_e, q_s, q_e, band);
}

// Wave-cooperative NW_score for deferred problems: lane n of a W-lane segment owns query column n and the segment
// sweeps the anti-diagonals, so t_len * q_len cells cost t_len + q_len - 1 steps. W = 64: one problem of 17..63
// columns per call; W = 16: four problems of 2..16 columns side by side. Same recurrences, tie rules and walk
// counters as nw_full; neighbours come by shuffles (cell (m, n+1) from lane n+1's last step, (m+1, n+1) from the
// step before that). All 64 lanes call this; segment g solves entry `first + g` (if < count) and its lane 0
// overwrites the entry with the result: e[2..6] = score, len, match, tGaps, qGaps.
template <int W>
__device__ void nw_coop(const Lane &L, const DevDB &db, const AlignArgs &A, int *qu, int first, int count, uint8_t *tbuf_wave) {
	constexpr int G = 64 / W;
	const int lane = threadIdx.x & 63, n = lane & (W - 1), g = lane / W;
	const bool live = first + g < count;
	int *e = qu + 1 + (live ? first + g : first) * QENT;
	uint8_t *tbuf = tbuf_wave + g * (TBUF / G);
	const int k = e[2], t_s = e[3], t_e = e[4], q_s = e[5], q_e = e[6], at = e[7];
	const int64_t rd = ((int64_t) e[9] << 32) | (uint32_t) e[8];
	const int tlen_total = db.tlen[at];
	const uint64_t *ts = db.tseq + db.tseq_off[at];
	int t_len = t_e - t_s;
	if(t_len < 0) t_len += tlen_total;
	const int q_len = q_e - q_s;
	const int U = L.U, W1 = L.W1;
	const uint32_t MA = 1u, TG = 1u << 10, QG = 1u << 20;
	const int low = (t_len + q_len) * (L.MM + U + W1);
	if(L.cnt && live && n == 0) atomicAdd(&L.cnt[5], (unsigned long long) t_len * q_len);
	// template bases of rows 0..t_len-1 (row m = position t_s + m, circular)
	if(live) for(int i = n; i < t_len; i += W) {
		int pos = t_s + i;
		if(pos >= tlen_total) pos -= tlen_total;
		tbuf[i] = (uint8_t) tn(ts, pos);
	}
	QView q;
	q.w = A.seq + A.seq_off[rd]; q.L = A.len[rd]; q.rc = e[10];
	q.N = A.N + A.N_off[rd]; q.nN = (int) (A.N_off[rd + 1] - A.N_off[rd]);
	const bool col = live && n < q_len;
	const int qc = col ? qn(q, q_s + n) : 0;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// state: `last` = the row this lane computed most recently (start: boundary row m = t_len), `before` = the row before it
	int lD, lP = low, lQ = low, bD = 0;
	uint32_t lTD, lTP = 0, lTQ = 0, bTD = 0;
	if(k == 2) { lD = 0; lTD = 0; } else { lD = W1 + (q_len - 1 - n) * U; lTD = TG * (uint32_t) max(0, q_len - n); }
	int best = low;
	uint32_t bestTD = 0;
	int steps = live ? t_len + q_len - 1 : 0;
	for(int o = 32; o > 0; o >>= 1) steps = max(steps, __shfl_xor(steps, o));
	for(int d = 0; d < steps; ++d) {
		int rD = __shfl_down(lD, 1, W), rQ = __shfl_down(lQ, 1, W), dD = __shfl_down(bD, 1, W);
		uint32_t rTD = __shfl_down(lTD, 1, W), rTQ = __shfl_down(lTQ, 1, W), dTD = __shfl_down(bTD, 1, W);
		const int i = d - (q_len - 1 - n);
		if(col && i >= 0 && i < t_len) {
			const int m = t_len - 1 - i;
			if(n == q_len - 1) {
				// boundary column q_len (nw.c:703-750, :757)
				rD = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
				rTD = (0 < k) ? 0u : QG * (uint32_t) (t_len - m);
				rTQ = 0; rQ = low;
				if(m + 1 == t_len) { dD = 0; dTD = 0; }
				else { dD = (0 < k) ? 0 : (W1 + (t_len - 2 - m) * U); dTD = (0 < k) ? 0u : QG * (uint32_t) (t_len - 1 - m); }
			}
			const int *drow = L.d + 5 * (int) tbuf[m];
			int Q = rD + W1, P = lD + W1, D, mv;
			bool ob = false;
			if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
			int x = rQ + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else ob = true;
			x = lP + U;
			if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else ob = true;
			x = dD + drow[qc];
			if(D <= x) { D = x; mv = 1; }
			const uint32_t TQ = TG + (ob ? rTD : rTQ);
			const uint32_t TP = QG + (ob ? lTD : lTP);
			const uint32_t TD = (mv == 1) ? (MA + dTD) : (mv >= 4 ? TP : TQ);
			bD = lD; bTD = lTD;
			lD = D; lP = P; lTD = TD; lTP = TP; lQ = Q; lTQ = TQ;
			if(n == 0 && k < 0 && best < D) { best = D; bestTD = TD; }
		}
	}
	// result selection (nw.c:830-845), per segment
	const int seg0 = g * W;
	int score;
	uint32_t st;
	if(k < 0) {
		score = __shfl(best, seg0); st = __shfl(bestTD, seg0);
	} else {
		score = __shfl(lD, seg0); st = __shfl(lTD, seg0);
	}
	{
		// k == -2: for n ascending: if(score <= D[0][n]) take it -> the largest n holding the row maximum, if it is >= score
		int mx = col ? lD : INT_MIN;
		for(int o = W / 2; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, W));
		const unsigned long long who = __ballot(col && lD == mx);
		const unsigned long long mine = (W == 64) ? who : ((who >> seg0) & ((1ull << (W & 63)) - 1ull));
		const int src = seg0 + (mine ? 63 - __clzll((long long) mine) : 0);
		const uint32_t stn = __shfl(lTD, src);
		if(k == -2 && mx >= score) { score = mx; st = stn; }
	}
	if(live && n == 0) {
		const Aln r = aln_from<10>(score, (uint64_t) st);
		e[2] = r.score; e[3] = r.len; e[4] = r.match; e[5] = r.tGaps; e[6] = r.qGaps;
	}
}

// The same for ONE problem of up to 64 * XC - 1 columns: lane n owns XC neighbouring columns (aligned to the right end, so
// only lane 0 can own fewer) and takes them right to left inside a step; between lanes the sweep is the anti-diagonal one
// of nw_coop -- the right neighbour (m, c+1) and the diagonal (m+1, c+1) of a lane's last column come from the first
// column of the lane to its right, one and two steps old.
// band > 0: NW_band_score (nw.c:892-1188) on the same sweep. In query coordinates the banded recurrences are the full ones
// restricted to the columns [c - band/2, c + band/2] of each row (c = (t_len + q_len)/2 at the last row, one less per row),
// with three differences: the leftmost cell of a row has no template-gap state (nw.c:1079-1105); where the band ends before
// the last column its right neighbour is a virtual cell (D = low, walk counter = one query gap more than the cell below
// it); the result is read off the leftmost cells. A column outside the band simply keeps its last value, which is exactly
// what the neighbouring column then needs (the cell below the virtual one, the diagonal of the band's last column).
template <int XW, bool banded>
__device__ void nw_coop_x(const Lane &L, const DevDB &db, const AlignArgs &A, int *qu, int ent, uint8_t *tbuf) {
	const int lane = threadIdx.x & 63;
	int *e = qu + 1 + ent * QENT;
	const int k = e[2], t_s = e[3], t_e = e[4], q_s = e[5], q_e = e[6], at = e[7];
	int band = banded ? (e[11] >> 1) : 0;
	const int64_t rd = ((int64_t) e[9] << 32) | (uint32_t) e[8];
	const int tlen_total = db.tlen[at];
	const uint64_t *ts = db.tseq + db.tseq_off[at];
	int t_len = t_e - t_s;
	if(t_len < 0) t_len += tlen_total;
	const int q_len = q_e - q_s;
	const int U = L.U, W1 = L.W1;
	const uint32_t MA = 1u, TG = 1u << 10, QG = 1u << 20;
	const int low = (t_len + q_len) * (L.MM + U + W1);
	if(band & 1) ++band;
	const int half = band >> 1;
	const int cbot = (t_len + q_len) >> 1;             // band centre of the last row (m = t_len - 1)
	if(L.cnt && lane == 0) atomicAdd(&L.cnt[5], (unsigned long long) t_len * (banded ? band + 1 : q_len));
	for(int i = lane; i < t_len; i += 64) {
		int pos = t_s + i;
		if(pos >= tlen_total) pos -= tlen_total;
		tbuf[i] = (uint8_t) tn(ts, pos);
	}
	QView q;
	q.w = A.seq + A.seq_off[rd]; q.L = A.len[rd]; q.rc = e[10];
	q.N = A.N + A.N_off[rd]; q.nN = (int) (A.N_off[rd + 1] - A.N_off[rd]);
	const int nl = (q_len + XW - 1) / XW;              // lanes in use
	const int c0 = q_len - (nl - lane) * XW;           // this lane's first column (negative: lane 0 owns fewer than XW)
	const bool act = lane < nl;
	int qc[XW], lD[XW], lP[XW], lQ[XW];
	uint32_t lTD[XW], lTP[XW], lTQ[XW];
#pragma unroll
	for(int j = 0; j < XW; ++j) {
		const int c = c0 + j;
		qc[j] = (act && c >= 0) ? qn(q, q_s + c) : 0;
		// boundary row m = t_len
		if(k == 2) { lD[j] = 0; lTD[j] = 0; } else { lD[j] = W1 + (q_len - 1 - c) * U; lTD[j] = TG * (uint32_t) max(0, q_len - c); }
		lP[j] = low; lQ[j] = low; lTP[j] = 0; lTQ[j] = 0;
	}
	int bD0 = 0;                                       // the row before `l` of this lane's first column, for the lane to the left
	uint32_t bTD0 = 0;
	int best = low, bm = 0;
	uint32_t bestTD = 0;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const int steps = t_len + nl - 1;
	for(int d = 0; d < steps; ++d) {
		// the first column of the lane to the right: its newest row (l) and the one before (b)
		const int nD = __shfl_down(lD[0], 1), nQ = __shfl_down(lQ[0], 1), nbD = __shfl_down(bD0, 1);
		const uint32_t nTD = __shfl_down(lTD[0], 1), nTQ = __shfl_down(lTQ[0], 1), nbTD = __shfl_down(bTD0, 1);
		const int i = d - (nl - 1 - lane);
		if(act && i >= 0 && i < t_len) {
			const int m = t_len - 1 - i;
			// columns of this row: [eq, sq]; `virt`: the band ends before the last column
			const int c = cbot - i;
			const int eq = banded ? max(c - half, 0) : -1;
			const bool virt = banded && c + half < q_len - 1;
			const int sq = virt ? c + half : q_len - 1;
			int rD, rQ, dD;
			uint32_t rTD, rTQ, dTD;
			// what the lane's last column sees to its right
			const int cr = c0 + XW;                         // first column of the lane to the right (q_len for the last lane)
			if(cr >= q_len) {
				// boundary column q_len (nw.c:703-750, :757)
				rD = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
				rTD = (0 < k) ? 0u : QG * (uint32_t) (t_len - m);
				rTQ = 0; rQ = low;
				if(m + 1 == t_len) { dD = 0; dTD = 0; }
				else { dD = (0 < k) ? 0 : (W1 + (t_len - 2 - m) * U); dTD = (0 < k) ? 0u : QG * (uint32_t) (t_len - 1 - m); }
			} else if(cr <= sq) { rD = nD; rQ = nQ; rTD = nTD; rTQ = nTQ; dD = nbD; dTD = nbTD; }   // it has done this row already
			else { rD = low; rQ = low; rTQ = 0; rTD = QG + nTD; dD = nD; dTD = nTD; }                    // outside the band: virtual cell, `l` is the row below
			const int *drow = L.d + 5 * (int) tbuf[m];
#pragma unroll
			for(int jj = 0; jj < XW; ++jj) {
				const int j = XW - 1 - jj;          // right to left; indices are compile-time after unrolling (arrays stay in registers)
				const int col = c0 + j;
				if(col >= 0 && col >= eq && col <= sq) {
					const int oD = lD[j];
					const uint32_t oTD = lTD[j];
					int D, Q, P;
					uint32_t TD, TQ, TP;
					if(col == eq) {
						// leftmost cell of the band (nw.c:1079-1105): no template-gap state
						int mv;
						bool ob = false;
						Q = rD + W1;
						const int x = rQ + U;
						if(Q < x) { Q = x; mv = 3; } else { mv = 2; ob = true; }
						D = dD + drow[qc[j]];
						if(Q <= D) mv = 1; else D = Q;
						TQ = TG + (ob ? rTD : rTQ);
						TD = (mv == 1) ? (MA + dTD) : TQ;
						P = low; TP = 0;
						if(eq == 0 && k < 0 && best < D) { best = D; bestTD = TD; bm = m; }
					} else {
						int mv;
						bool ob = false;
						Q = rD + W1; P = oD + W1;
						if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
						int x = rQ + U;
						if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else ob = true;
						x = lP[j] + U;
						if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else ob = true;
						x = dD + drow[qc[j]];
						if(D <= x) { D = x; mv = 1; }
						TQ = TG + (ob ? rTD : rTQ);
						TP = QG + (ob ? oTD : lTP[j]);
						TD = (mv == 1) ? (MA + dTD) : (mv >= 4 ? TP : TQ);
						if(!banded && col == 0 && k < 0 && best < D) { best = D; bestTD = TD; }
					}
					if(j == 0) { bD0 = oD; bTD0 = oTD; }
					lD[j] = D; lP[j] = P; lTD[j] = TD; lTP[j] = TP; lQ[j] = Q; lTQ[j] = TQ;
					// for the column to the left: this cell is its right neighbour, the cell below it its diagonal
					rD = D; rQ = Q; rTD = TD; rTQ = TQ; dD = oD; dTD = oTD;
				} else if(col >= 0 && col == sq + 1) {
					// first column outside the band: the virtual cell for the band's last column
					rD = low; rQ = low; rTQ = 0; rTD = QG + lTD[j]; dD = lD[j]; dTD = lTD[j];
				}
			}
		}
	}
	// result selection. Full: nw.c:830-845, banded: nw.c:1148-1170. Column `cres` (0, or the leftmost column of the last
	// row's band) holds the final value; lane 0 tracked the best leftmost cell for k < 0.
	const int cfin = cbot - (t_len - 1);
	const int cres = banded ? max(cfin - half, 0) : 0;
	const int sfin = (banded && cfin + half < q_len - 1) ? cfin + half : q_len - 1;
	const int owner = nl - 1 - (q_len - 1 - cres) / XW;
	int Dres = 0;
	uint32_t TDres = 0;
#pragma unroll
	for(int j = 0; j < XW; ++j) if(c0 + j == cres) { Dres = lD[j]; TDres = lTD[j]; }
	Dres = __shfl(Dres, owner); TDres = __shfl(TDres, owner);
	// the best cell was tracked by the owner of column 0 (lane 0)
	int score = __shfl(best, 0);
	uint32_t st = __shfl(bestTD, 0);
	const int bm0 = __shfl(bm, 0);
	if(banded) { if(bm0 == 0) { score = Dres; st = TDres; } }
	else if(!(k < 0)) { score = Dres; st = TDres; }
	if(k == -2) {
		// for n ascending: if(score <= D[0][n]) take it -> the largest column holding the row maximum, if it is >= score
		int mx = INT_MIN, mc = -1;
		uint32_t mtd = 0;
#pragma unroll
		for(int j = 0; j < XW; ++j) if(act && c0 + j >= cres && c0 + j <= sfin && lD[j] >= mx) { mx = lD[j]; mc = c0 + j; mtd = lTD[j]; }
		int wmx = mx;
		for(int o = 32; o > 0; o >>= 1) wmx = max(wmx, __shfl_xor(wmx, o));
		const unsigned long long who = __ballot(act && mc >= 0 && mx == wmx);
		const int src = who ? 63 - __clzll((long long) who) : 0;
		const uint32_t stn = __shfl(mtd, src);
		if(wmx >= score) { score = wmx; st = stn; }
	}
	if(lane == 0) {
		const Aln r = aln_from<10>(score, (uint64_t) st);
		e[2] = r.score; e[3] = r.len; e[4] = r.match; e[5] = r.tGaps; e[6] = r.qGaps;
	}
}

// chainSeeds, chain.c:79-260; returns best start, *bestScore its score
__device__ int chain_seeds(const Lane &L, int n, int q_len, int t_len, int k, unsigned *mapQ) {
	const int W1 = L.W1, U = L.U, M = L.M, MM = L.MM;
	int best = 0, second = 0, bestPos = n - 1;
	MEMA(L, 5, n) = 0; MEMA(L, 6, n) = 0;
	for(int i = n - 1; i >= 0; --i) {
		const int wi = MEMA(L, 4, i);
		const int weight = wi * M, tEnd = MEMA(L, 1, i), qEnd = MEMA(L, 3, i);
		int nxt = 0;
		int span = min(t_len - tEnd, q_len - qEnd);
		int gap = span - 1;
		gap = gap ? gap * U + W1 : W1;
		int sub = mism_score(span, k, M, MM);
		int score = weight + (sub < gap ? gap : sub);
		const int lim = min(n, i + 128);
		for(int j = i + 1; j < lim; ++j) {
			const int qSj = MEMA(L, 2, j), tSj = MEMA(L, 0, j);
			if(qEnd < qSj) {
				if(tEnd < tSj) {
					const int tGap = tSj - tEnd, qGap = qSj - qEnd;
					int g = abs(tGap - qGap);
					if(g) g = (g - 1) * U + W1;
					g += weight + MEMA(L, 5, j) + mism_score(min(tGap, qGap), k, M, MM);
					if(score <= g) { score = g; nxt = j; }
				} else if(k <= MEMA(L, 1, j) - tEnd) {
					int g = qSj - qEnd;
					if(g) g = (g - 1) * U + W1;
					g += weight + MEMA(L, 5, j) - (tSj - tEnd) * M;
					if(score < g) { score = g; nxt = j; }
				}
			} else if(k <= MEMA(L, 3, j) - qEnd) {
				const int tStart = tSj + qEnd - qSj;
				if(tEnd < tStart) {
					int g = tStart - tEnd;
					if(g) g = (g - 1) * U + W1;
					g += weight + MEMA(L, 5, j) - (tStart - tEnd) * M;
					if(score < g) { score = g; nxt = j; }
				}
			}
		}
		MEMA(L, 6, i) = nxt;
		MEMA(L, 4, i) = nxt ? (wi + MEMA(L, 4, nxt) - k + 1) : (wi - (k - 1));
		MEMA(L, 5, i) = score;
		span = min(MEMA(L, 0, i), MEMA(L, 2, i));
		gap = span - 1;
		if(0 < gap) gap = gap * U + W1; else if(gap == 0) gap = W1; else gap = 0;
		sub = mism_score(span, k, M, MM);
		score += sub < gap ? gap : sub;
		if(best <= score) {
			if(nxt != bestPos) second = best;
			best = score; bestPos = i;
		} else if(second <= score && nxt != bestPos) {
			second = best;
		}
	}
	if(0 < best) {
		const double wq = fmin(1.0, MEMA(L, 4, bestPos) / 10.0);
		*mapQ = (unsigned) ceil(40 * (1 - 1.0 * second / best) * wq * log((double) best));
	} else *mapQ = 0;
	MEMA(L, 5, bestPos) = best;
	return bestPos;
}

// one maximal exact match through k-mer hit (q pos j, template 1-based pos1); returns its query end.
// Same result as the reference's base-by-base loops (align.c:548-575), done 32 bases per step on the
// 2-bit words: lowq = first query position after the previous N (an N never matches, so the backward
// walk cannot cross it), segstop = end of the N-free segment.
__device__ __forceinline__ void mem_extend(const uint64_t *ts, int t_len, const QView &q, int j, int pos1, int k, int lowq, int segstop,
                                           int &tS, int &tE, int &qS, int &qE) {
	int kk = j - 1, prev = pos1 - 2;
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
	qS = kk + 1; tS = prev + 2;
	int value = pos1 + k - 1, l = j + k;
	for(;;) {
		const int room = min(segstop - l, t_len - value);
		if(room <= 0) break;
		const int step = min(32, room);
		const uint64_t x = (qwin(q, l) ^ win2(ts, value)) >> (64 - 2 * step);
		const int same = x ? (__clzll((long long) x) - (64 - 2 * step)) >> 1 : step;
		l += same; value += same;
		if(same < step) break;
	}
	qE = l; tE = value + 1;
}

__device__ int add_mem(const Lane &L, int m, const uint64_t *ts, int t_len, const QView &q, int j, int pos1, int k, int lowq, int segstop) {
	// backward
	int kk = j - 1, prev = pos1 - 2;
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
	MEMA(L, 2, m) = kk + 1; MEMA(L, 0, m) = prev + 2;
	// forward
	int value = pos1 + k - 1, l = j + k;
	for(;;) {
		const int room = min(segstop - l, t_len - value);
		if(room <= 0) break;
		const int step = min(32, room);
		const uint64_t x = (qwin(q, l) ^ win2(ts, value)) >> (64 - 2 * step);
		const int same = x ? (__clzll((long long) x) - (64 - 2 * step)) >> 1 : step;
		l += same; value += same;
		if(same < step) break;
	}
	MEMA(L, 3, m) = l; MEMA(L, 1, m) = value + 1;
	MEMA(L, 4, m) = l - (kk + 1);
	if(L.cnt) atomicAdd(&L.cnt[4], (unsigned long long) (l - (kk + 1)));
	return l;
}

// anker_rc_comp, align.c:993-1176: a read whose two strands tied in stage 2 is seeded on both strands of
// the template; the strand with the larger MEM coverage wins (forward on equality) and its MEMs are
// handed to KMA_score. Returns +n (forward, n MEMs at [0,n)), -n (reverse) or 0. Rare path: base by base.
__device__ int anker_rc_comp(const Lane &L, const DevDB &db, int t, const uint64_t *ts, int t_len, const QView &qf, int *status) {
	const int k = (int) db.kmersize, q_len = qf.L, cap = L.cap1 - 1;
	QView qr = qf; qr.rc = qf.rc ^ 1;
	int bestScore = 0, score = 0, score_r = 0, mem_count = 0, tot = 0, plen = 0;
	// query bounds (align.c:1029-1044): the forward pass starts at q_start when there is one (no preseed then) and both passes
	// stop looking for seeds at q_end -- the reverse pass with the bounds counted from the other end; the stretches themselves
	// still end at the N's (or the read end), not at the bound
	const int fb0 = qf.b0, fb1 = qb1(qf);
	for(int rc = 0; rc < 2; ++rc) {
		const QView &q = rc ? qr : qf;
		int i = 0;
		const int stop = rc ? q_len - fb0 : fb1;
		if(rc) { score = score_r; plen = mem_count; i = q_len - fb1; }
		else if(fb0) i = fb0;
		else {
			// preseed (align.c:750-768): every k-th k-mer, built from byte codes (N = 4), past-the-end bytes = 0
			bool hit = false;
			const int span = fb1 - fb0;
			for(i = 0; i < span && !hit; i += k) {
				uint64_t key = 0;
				for(int x = 0; x < k; ++x) key = (x ? (key << 2) : 0ull) | (uint64_t) ((i + x < q_len) ? qn(q, i + x) : 0);
				if(key <= 0xFFFFFFFFull && tpos_get(db, t, (uint32_t) key) != 0) hit = true;
			}
			if(hit) i = 0;
		}
		score_r = 0; mem_count = 0;
		int ni = 0;
		while(i < stop) {
			const int end = qN_at(q, ++ni) - k + 1;
			while(i < end) {
				const int v = tpos_get(db, t, q_kmer(q, i, k));
				if(v == 0) { ++i; continue; }
				if(v > 0) {
					if(tot >= cap) { *status = 1; return 0; }
					int value = v, prev = value - 2, j;
					for(j = i - 1; 0 <= j && 0 <= prev && qn(q, j) == tn(ts, prev); --j) { --prev; ++score_r; }
					MEMA(L, 2, tot) = j + 1; MEMA(L, 0, tot) = prev + 2;
					value += k - 1; i += k; score_r += k;
					while(i < end && value < t_len && qn(q, i) == tn(ts, value)) { ++i; ++value; ++score_r; }
					MEMA(L, 3, tot) = i; MEMA(L, 1, tot) = value + 1;
					MEMA(L, 4, tot) = (value + 1) - (prev + 2);
					++mem_count; ++tot;
					++i;
				} else {
					score_r += k;
					const int32_t *dl = db.tpos_dups + (-v - 1);
					const int cnt = dl[0];
					int bias = i;
					for(int c = 1; c <= cnt; ++c) {
						if(tot >= cap) { *status = 1; return 0; }
						int value = dl[c], prev = value - 2, j, kk = i;
						for(j = kk - 1; 0 <= j && 0 <= prev && qn(q, j) == tn(ts, prev); --j) --prev;
						MEMA(L, 2, tot) = j + 1; MEMA(L, 0, tot) = prev + 2;
						value += k - 1; kk += k;
						while(kk < end && value < t_len && qn(q, kk) == tn(ts, value)) { ++kk; ++value; }
						MEMA(L, 3, tot) = kk; MEMA(L, 1, tot) = value + 1;
						MEMA(L, 4, tot) = kk - (j + 1);
						++mem_count; ++tot;
						bias = max(bias, kk);
					}
					score_r += bias - i;
					i = bias + 1;
				}
			}
			i = end + k;
		}
		bestScore = max(bestScore, score_r);
	}
	// one2one is set under -1t1 (kma.c:686-688, 1429)
	if(bestScore < k && bestScore * k < (q_len - k - bestScore)) return 0;
	if(bestScore == score) return plen;          // forward (also on equality); plen > 0 whenever bestScore > 0
	for(int x = 0; x < mem_count; ++x) {
		for(int a = 0; a < 5; ++a) MEMA(L, a, x) = MEMA(L, a, plen + x);
	}
	return -mem_count;
}

// KMA_score, align.c:509-748. status: 0 ok, 1 = MEM capacity exceeded
__device__ Aln kma_score(const Lane &L, const DevDB &db, int t, const uint64_t *ts, int t_len, const QView &q, int mq, int preseeded, int *status) {
	const Aln FAIL = {0, 1, 0, 0, 0, 0};
	const int k = (int) db.kmersize, q_len = q.L, bw = 64, cap = L.cap1 - 1;
	int nm = 0, j = q.b0, lowq = 0;
#ifdef KMAHIP_DIAG
	if(L.ablate & 2) return FAIL;
#endif
	// MEMs left by anker_rc_comp are used as they are (align.c:530-532)
	if(preseeded > 0) nm = preseeded;
	for(int i = 1; preseeded <= 0 && i <= q.nN + 1; ++i) {
		const int Ni = qN_at(q, i);
		const int end = (i != q.nN + 1) ? Ni - k + 1 : qb1(q) - k + 1;
		const int segstop = end + k - 1;
		while(j < end) {
			// two k-mer starts per step, both gathers in flight together: the usual miss (the k-mer that starts on a
			// mismatch) is followed by a hit one base later, and a wave waits for its slowest lane's step count
			int v, v2;
			uint32_t km1, km2;
			q_kmer2(q, j, k, km1, km2);
			tpos_get2(db, t, km1, km2, j + 1 < end, v, v2);
			if(L.cnt) atomicAdd(&L.cnt[3], 1ull);
			if(v == 0) {
				if(j + 1 < end && L.cnt) atomicAdd(&L.cnt[3], 1ull);
				if(v2 == 0) { j += 2; continue; }
				++j; v = v2;
			}
			if(v > 0) {
				if(nm >= cap) { *status = 1; return FAIL; }
				j = add_mem(L, nm, ts, t_len, q, j, v, k, lowq, segstop);
				++nm;
			} else {
				const int32_t *dl = db.tpos_dups + (-v - 1);
				const int cnt = dl[0];
				int bias = j;
				for(int c = 1; c <= cnt; ++c) {
					if(nm >= cap) { *status = 1; return FAIL; }
					const int qe = add_mem(L, nm, ts, t_len, q, j, dl[c], k, lowq, segstop);
					++nm;
					bias = max(bias, qe);
				}
				j = bias + 1;
			}
		}
		j = Ni + 1;
		lowq = Ni + 1;
	}
	if(!nm) return FAIL;
	unsigned mapQ = 0;
#ifdef KMAHIP_DIAG
	if(L.ablate & 4) return FAIL;
#endif
	int start = chain_seeds(L, nm, q_len, t_len, k, &mapQ);
	if(mapQ < (unsigned) mq || MEMA(L, 5, start) < k) return FAIL;

	// leading tail (leadTailAln, align.c:53-131)
	Aln S = {0, 0, 0, 0, 0, 0};
	{
		const int t_e = MEMA(L, 0, start) - 1, q_e = MEMA(L, 2, start);
		S.pos = t_e;
		if(q_e) {
			int t_s = 0, q_s = 0;
			if((q_e << 1) < t_e || (q_e + bw) < t_e) t_s = t_e - (q_e + (q_e < bw ? q_e : bw));
			else if((t_e << 1) < q_e || (t_e + bw) < q_e) q_s = q_e - (t_e + (t_e < bw ? t_e : bw));
			if(t_e - t_s > 0 && q_e - q_s > 0) {
				const int band = abs(t_e - t_s - q_e + q_s) + bw;
				const Aln r = nw_auto(L, ts, t_len, q, -1 - (t_s == 0), t_s, t_e, q_s, q_e, t_e - t_s, band);
				S.pos -= r.len - r.tGaps;
				S.score = r.score; S.len = r.len; S.match = r.match; S.tGaps = r.tGaps; S.qGaps = r.qGaps;
			}
		}
	}
	for(;;) {
		const int qS = MEMA(L, 2, start), qE = MEMA(L, 3, start);
		S.len += qE - qS; S.match += qE - qS;
		if(L.diag_uniform) S.score += (qE - qS) * L.d[0];
		else for(int i = qS; i < qE; ++i) { const int b = qn(q, i); S.score += L.d[6 * b]; }
		const int nxt = MEMA(L, 6, start);
		if(!nxt) break;
		const int q_s = qE, t_s = MEMA(L, 1, start) - 1;
		start = nxt;
		int qSn = MEMA(L, 2, start), tSn = MEMA(L, 0, start);
		if(qSn < q_s) { tSn += q_s - qSn; qSn = q_s; }
		int t_e = tSn - 1, t_l;
		if(t_e < t_s) {
			if(t_s <= MEMA(L, 1, start)) { qSn += t_s - t_e; t_e = t_s; t_l = 0; }
			else t_l = t_len - t_s + t_e;
		} else t_l = t_e - t_s;
		MEMA(L, 2, start) = qSn; MEMA(L, 0, start) = tSn;
		const int q_e = qSn;
		if(abs(t_l - q_e + q_s) * L.U > q_len * L.M || t_l > q_len || q_e - q_s > (q_len >> 1)) return FAIL;
		if(t_l > 0 || q_e - q_s > 0) {
			const int band = abs(t_l - q_e + q_s) + bw;
			const Aln r = nw_auto(L, ts, t_len, q, 0, t_s, t_e, q_s, q_e, t_l, band);
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
	}
	{	// trailing tail (trailTailAln, align.c:140-212)
		const int t_s = MEMA(L, 1, start) - 1, q_s = MEMA(L, 3, start);
		int q_e = q_len, t_e = t_len;
		if(((q_len - q_s) << 1) < (t_len - t_s) || (q_len - q_s + bw) < (t_len - t_s)) {
			t_e = q_len - q_s; t_e = t_s + (t_e + (t_e < bw ? t_e : bw));
		} else if(((t_len - t_s) << 1) < (q_len - q_s) || (t_len - t_s + bw) < (q_len - q_s)) {
			q_e = t_len - t_s; q_e = q_s + (q_e + (q_e < bw ? q_e : bw));
		}
		if(t_e - t_s > 0 && q_e - q_s > 0) {
			const int band = abs(t_e - t_s - q_e + q_s) + bw;
			const Aln r = nw_auto(L, ts, t_len, q, 1 + (t_e == t_len), t_s, t_e, q_s, q_e, t_e - t_s, band);
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
	}
	return S;
}

// STATS = work-counter launches (atomics in the hot loops; never timed) get their own symbol so that profiler
// averages of the production kernel stay clean
// ---- seeding on its own: the MEM search of KMA_score (align.c:534-638) needs a fraction of the registers of what follows
// it, and all it does is wait for dependent gathers -- so it runs as its own kernel at 8 waves / SIMD (twice the gathers in
// flight), one lane per single-end task, and hands up to SEEDS MEMs per task to the main kernel. Tasks it cannot serve
// (strand ties, more MEMs, reads >= 64 k bases) are marked -1 and seeded by the main kernel as before.
// Long reads (single end, strand decided by stage 2, no N's): a lane would walk hundreds of MEMs and DP problems of such a read alone;
// their tasks go through the wavefront-per-read pipeline of longtrace.hip in KMA_score mode instead (launch_align: long_tasks)
__device__ __forceinline__ bool long_routed(const AlignArgs &A, int64_t r) {
	return A.long_min > 0 && A.len[r] >= A.long_min && A.rc_flag[r] > 0 && A.N_off[r + 1] == A.N_off[r];
}

// the MEM search of one oriented read against one template; returns the number of MEMs (0..SEEDS) or -1 (left to the main kernel)
__device__ __forceinline__ int seed_view(const AlignArgs &A, int t, const QView &q, int k, uint2 *mem) {
	const int L = q.L;
	if(L < k || L > 0xFFFF) return -1;
	const uint4 ma = A.db.tmeta[2 * (size_t) t], mb = A.db.tmeta[2 * (size_t) t + 1];
	const int t_len = (int) mb.x;
	const uint64_t *ts = A.db.tseq + (((uint64_t) ma.y << 32) | ma.x);
	const uint2 *tab = A.db.tpos_slots + (((uint64_t) ma.w << 32) | ma.z);
	const uint32_t tsh = mb.y;
	int nm = 0, j = q.b0, lowq = 0;
	for(int i = 1; i <= q.nN + 1; ++i) {
		const int Ni = qN_at(q, i);
		const int end = (i != q.nN + 1) ? Ni - k + 1 : qb1(q) - k + 1;
		const int segstop = end + k - 1;
		while(j < end) {
			int v, v2;
			uint32_t km1, km2;
			q_kmer2(q, j, k, km1, km2);
			tpos_get2(tab, tsh, km1, km2, j + 1 < end, v, v2);
			if(v == 0) {
				if(v2 == 0) { j += 2; continue; }
				++j; v = v2;
			}
			if(v < 0 || nm >= SEEDS) return -1;      // duplicated k-mer (several MEMs per lookup) or too many MEMs
			int tS, tE, qS, qE;
			mem_extend(ts, t_len, q, j, v, k, lowq, segstop, tS, tE, qS, qE);
#pragma unroll
			for(int x = 0; x < SEEDS; ++x) if(x == nm) mem[x] = make_uint2((uint32_t) tS, (uint32_t) qS | ((uint32_t) (qE - qS) << 16));
			++nm;
			j = qE;
		}
		j = Ni + 1;
		lowq = Ni + 1;
	}
	return nm;
}

__global__ __launch_bounds__(256, 8) void seed_tasks_kernel(const AlignArgs A) {
	const int64_t n_tasks = A.T_off[A.n_reads];
	if(n_tasks > A.tasks_cap) return;
	const int k = (int) A.db.kmersize;
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	const int slots = A.seed_slots;
	for(int64_t task = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; task < n_tasks; task += stride) {
		const int64_t r = A.t_rec[task];
		const int t = abs(A.T[task]);
		// the reads this task aligns and their orientation: exactly the choices of align_tasks_kernel's phase A
		int views = 0, rcstate = 0;
		int64_t rd0 = r;
		int rc0 = 0;
		if(!A.pe_mode) {
			if(A.rc_flag[r] > 0 && !long_routed(A, r)) { views = 1; rc0 = (A.flag[r] & 16) ? 1 : 0; }
		} else {
			const int64_t p0 = r & ~1ll;
			const bool couple = (r & 1) && A.rec_mate[p0] >= 0 && A.rec_mate[r] >= 0 && A.T_off[p0 + 1] == A.T_off[p0];
			if(couple) {
				views = 2;
				for(int64_t j = A.T_off[r]; j <= task; ++j) if(A.T[j] < 0) { rcstate = 1; break; }
			} else if(A.rc_flag[r] > 0 && A.rec_mate[r] >= 0) { views = 1; rd0 = p0 + A.rec_mate[r]; rc0 = A.rec_rc[r]; }
		}
		for(int m = 0; m < slots; ++m) {
			int nm = -1;
			uint2 mem[SEEDS];
			if(m < views) {
				int64_t rd = rd0;
				int rc = rc0;
				if(views == 2) { const int64_t rec = (r & ~1ll) + m; rd = (r & ~1ll) + A.rec_mate[rec]; rc = A.rec_rc[rec] ^ rcstate; }
				QView q;
				q.w = A.seq + A.seq_off[rd]; q.L = A.len[rd]; q.rc = rc;
				q_set_bounds(q, A.q_start, A.q_end, rd);
				q.N = A.N + A.N_off[rd]; q.nN = (int) (A.N_off[rd + 1] - A.N_off[rd]);
				nm = seed_view(A, t, q, k, mem);
			}
			A.seed_n[task * slots + m] = nm;
#pragma unroll
			for(int x = 0; x < SEEDS; ++x) if(x < nm) A.seed_mem[(task * slots + m) * SEEDS + x] = mem[x];
		}
	}
}

// phase C of a task: the filters of alnFragsSE (alnfrags.c:1127-1168) / the per-record part of alnFragsPenaltyPE (:1630-1775) on the
// sums of KMA_score, into the task columns the per-read reduction reads. Shared by the general and the register-only kernel.
template <bool PEM>
__device__ __forceinline__ void task_finish(const AlignArgs &A, int64_t task, int kind, const Aln &S0, const Aln &S1, int tmpl_out, int t_len,
                                            int qlen0, int qlen1, int k) {
	if(PEM && kind == 2) {
		int bt = 0, btr = 0, bs = -1, be = -1, raw_b = 0;
		for(int m = 0; m < 2; ++m) {
			const Aln &st = m ? S1 : S0;
			const int ql = m ? qlen1 : qlen0;
			int sc = st.score, s0 = 0, e0 = 0;
			double nrm = 0.0;
			if(A.minlen <= st.len && 0 < sc && ((A.mrc * ql <= st.len - st.qGaps) || (A.mrc * t_len <= st.len - st.tGaps))) {
				s0 = st.pos; e0 = st.pos + st.len - st.tGaps;
				if(s0 == 0) sc += A.Wl;
				if(e0 == t_len) sc += A.Wl;
				nrm = 1.0 * sc / st.len;
			} else sc = 0;
			const bool ok = sc > k && nrm >= A.scoreT;
			if(m == 0) {
				if(ok) { bt = sc; bs = s0; be = e0; }
			} else {
				if(ok) {
					btr = sc;
					if(bt) { if(s0 < bs) bs = s0; else be = e0; }
					else { bs = s0; be = e0; }
				}
				raw_b = sc;
			}
		}
		A.t_tmpl[task] = raw_b;      // couples: raw second-record score (joins the first in compScore, :1771)
		A.t_score[task] = bt; A.t_alen[task] = btr; A.t_start[task] = bs; A.t_end[task] = be; A.t_norm[task] = 0.0;
		return;
	}
	int rs = 0, alen = 0, start = 0, end = 0;
	double norm = 0.0;
	if(kind == 1) {
		// alnFragsSE, alnfrags.c:1127-1168
		const Aln &st = S0;
		const int q_len = qlen0;
		alen = st.len; start = st.pos;
		end = start + alen - st.tGaps;
		if(t_len < end) end -= t_len;
		const double denom = (q_len <= alen || t_len <= alen) ? (double) alen : (double) min(q_len, t_len);
		rs = st.score;
		if(A.minlen <= alen && ((A.mrc * q_len <= st.len - st.qGaps) || (A.mrc * t_len <= st.len - st.tGaps))) norm = rs / denom;
		else { rs = 0; norm = 0.0; }
	}
	A.t_tmpl[task] = tmpl_out;
	A.t_score[task] = rs; A.t_alen[task] = alen; A.t_start[task] = start; A.t_end[task] = end; A.t_norm[task] = norm;
}

// PEM: paired records (couples, mate / orientation tables); the single-end instantiation carries none of that code
template <bool STATS, bool PEM>
__global__ __launch_bounds__(ATHREADS, 4) void align_tasks_kernel(const AlignArgs A) {
	__shared__ int s_d[25];
	__shared__ uint32_t s_wide[(ATHREADS / 64) * WSLOTS * 4 * WCOLS];
	__shared__ int s_queue[(ATHREADS / 64) * QINTS];
	__shared__ int s_xcnt[(ATHREADS / 64) * 4];
	__shared__ uint8_t s_tbuf[(ATHREADS / 64) * TBUF];
	if(threadIdx.x < 25) s_d[threadIdx.x] = A.d[threadIdx.x];
	__syncthreads();
	const int64_t gtid = (int64_t) blockIdx.x * ATHREADS + threadIdx.x;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	Lane L;
	L.s32 = A.s32 + gtid; L.s64 = A.s64 + gtid; L.lanes = A.lanes; L.cap1 = A.mem_cap + 1; L.ncols = A.ncols;
	L.r32 = A.s32 + (int64_t) 7 * (A.mem_cap + 1) * A.lanes + gtid * (2 * (int64_t) A.ncols);
	L.r64 = A.s64 + gtid * (2 * (int64_t) A.ncols);
	L.d = s_d; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1;
	L.cnt = STATS ? A.counters : nullptr;
	L.wide = s_wide + wave * WSLOTS * 4 * WCOLS;
	L.queue = s_queue + wave * QINTS;     // wide queue, then the narrow queue, then the tiny queue
	L.xq_cap = A.xq_cap; L.xq_cnt = s_xcnt + wave * 4;
	L.xq = A.xq_cap ? A.xq + (size_t) (gtid >> 6) * 4 * A.xq_cap * QENT : nullptr;
	L.q_at = 0; L.q_mate = 0; L.q_rd = 0;
	L.ablate = A.ablate; L.gap_m_max = A.gap_m_max;
#ifdef KMAHIP_DIAG
	if(A.ablate & 64) L.queue = nullptr;      // ablation: no cooperative DP
#endif
	int *const queue = s_queue + wave * QINTS;
	int *const queueN = queue + (1 + QCAP * QENT);
	int *const queueT = queueN + (1 + QCAPN * QENT);
	int *const queueX = queueT + (1 + QCAPT * QENT);
	uint8_t *const tbuf = s_tbuf + wave * TBUF;
	L.diag_uniform = (s_d[0] == s_d[6] && s_d[0] == s_d[12] && s_d[0] == s_d[18]);
	const int64_t n_all = A.T_off[A.n_reads];
	if(n_all > A.tasks_cap) return;
	// (run over a list: the tasks the register-only kernel handed on; its launch has ended, the count is final)
	const int64_t n_tasks = A.use_list ? (int64_t) A.counters[2] : n_all;
	unsigned long long *const hand_out = &A.counters[A.use_list ? 8 : 7];
	const int k = (int) A.db.kmersize;
	// the wave stays together: every round each lane does its own task up to the (deferred) wide DP problems,
	// then all 64 lanes solve those, then each lane finishes its task
	// Tasks are handed out 64 at a time from a device-wide counter: with a fixed stride the wave that meets a rare long DP
	// (a 100-column tail walked by one lane takes milliseconds) would still own its share of the remaining tasks.
	for(;;) {
		unsigned long long base = 0;
		if(lane == 0) base = atomicAdd(hand_out, (unsigned long long) A.per_round);
		base = __shfl(base, 0);
		if((int64_t) base >= n_tasks) break;
		const bool have = lane < A.per_round && (int64_t) base + lane < n_tasks;
		const int64_t task = !A.use_list ? (int64_t) base + lane : (have ? A.slow_list[(int64_t) base + lane] : 0);
		if(lane == 0) { queue[0] = 0; queueN[0] = 0; queueT[0] = 0; queueX[0] = 0; }
		if(lane < 4) L.xq_cnt[lane] = 0;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// ---- phase A -------------------------------------------------------------------------------
#ifdef KMAHIP_DIAG
		const unsigned long long dbg_t0 = wall_clock64();
#endif
		int kind = 0;                  // 0 nothing, 1 single record, 2 couple
		Aln S0 = {0, 0, 0, 0, 0, 0}, S1 = {0, 1, 0, 0, 0, 0};
		int tmpl_out = 0, t_len = 0, qlen0 = 0, qlen1 = 0;
		if(have) {
			const int64_t r = A.t_rec[task];
			tmpl_out = A.T[task];
			int64_t rd = r;
			int orient = (A.flag[r] & 16) ? 1 : 0;
			bool couple = false;
			if(PEM) {
				const int64_t p0 = r & ~1ll;
				rd = p0 + max(0, A.rec_mate[r]);
				orient = A.rec_rc[r];
				couple = (r & 1) && A.rec_mate[p0] >= 0 && A.rec_mate[r] >= 0 && A.T_off[p0 + 1] == A.T_off[p0];
			}
			const int rcf = A.rc_flag[r];
			const int at = abs(tmpl_out);
			t_len = A.db.tlen[at];
			const uint64_t *ts = A.db.tseq + A.db.tseq_off[at];
			int status = 0;
			L.q_at = at;
			if(!PEM && long_routed(A, r)) kind = 3;          // (scored by the long-read pipeline)
			else if(PEM && couple) {
				// alnFragsPenaltyPE, alnfrags.c:1630-1775: both records of the pair against this candidate. Both are
				// flipped once the list reaches its first negative id and stay flipped (:1633-1647).
				kind = 2;
				int rcstate = 0;
				for(int64_t j = A.T_off[r]; j <= task; ++j) if(A.T[j] < 0) { rcstate = 1; break; }
				if(L.cnt) atomicAdd(&L.cnt[6], 1ull);
				for(int m = 0; m < 2; ++m) {
					const int64_t rec = (r & ~1ll) + m;
					const int64_t rdm = (r & ~1ll) + A.rec_mate[rec];
					QView q;
					q.w = A.seq + A.seq_off[rdm]; q.L = A.len[rdm]; q.rc = A.rec_rc[rec] ^ rcstate;
					q.N = A.N + A.N_off[rdm]; q.nN = (int) (A.N_off[rdm + 1] - A.N_off[rdm]);
					L.q_mate = m; L.q_rd = rdm;
					Aln st = {0, 1, 0, 0, 0, 0};
					if(q.L >= k) {
						const int pre = (!STATS && A.seed_n) ? A.seed_n[task * 2 + m] : -1;
						if(pre > 0) {
							for(int x = 0; x < pre; ++x) {
								const uint2 e = A.seed_mem[(task * 2 + m) * SEEDS + x];
								const int qS = (int) (e.y & 0xFFFFu), ln = (int) (e.y >> 16);
								MEMA(L, 0, x) = (int) e.x; MEMA(L, 1, x) = (int) e.x + ln; MEMA(L, 2, x) = qS; MEMA(L, 3, x) = qS + ln; MEMA(L, 4, x) = ln;
							}
							st = kma_score(L, A.db, at, ts, t_len, q, A.mq, pre, &status);
						} else if(pre < 0) st = kma_score(L, A.db, at, ts, t_len, q, A.mq, 0, &status);
					}
					if(m == 0) { S0 = st; qlen0 = q.L; } else { S1 = st; qlen1 = q.L; }
				}
			} else if(rcf != 0 && A.len[rd] >= k && (!PEM || A.rec_mate[r] >= 0)) {
				kind = 1;
				QView q;
				q.w = A.seq + A.seq_off[rd]; q.L = A.len[rd]; q.rc = orient;
				q_set_bounds(q, A.q_start, A.q_end, rd);
				q.N = A.N + A.N_off[rd]; q.nN = (int) (A.N_off[rd + 1] - A.N_off[rd]);
				qlen0 = q.L;
				L.q_mate = 0; L.q_rd = rd;
				if(L.cnt) atomicAdd(&L.cnt[6], 1ull);
				if(rcf > 0) {
					// MEMs found by seed_tasks_kernel (the work-counting launch seeds here so that its counters are complete)
					const int pre = (!STATS && A.seed_n) ? A.seed_n[task * (PEM ? 2 : 1)] : -1;
					if(pre > 0) {
						for(int m = 0; m < pre; ++m) {
							const uint2 e = A.seed_mem[task * (PEM ? 2 : 1) * SEEDS + m];
							const int qS = (int) (e.y & 0xFFFFu), ln = (int) (e.y >> 16);
							MEMA(L, 0, m) = (int) e.x; MEMA(L, 1, m) = (int) e.x + ln; MEMA(L, 2, m) = qS; MEMA(L, 3, m) = qS + ln; MEMA(L, 4, m) = ln;
						}
						S0 = kma_score(L, A.db, at, ts, t_len, q, A.mq, pre, &status);
					} else if(pre == 0) S0 = Aln{0, 1, 0, 0, 0, 0};
					else S0 = kma_score(L, A.db, at, ts, t_len, q, A.mq, 0, &status);
				} else {
					// strand tie: per template strand decision (alnfrags.c:1101-1124)
					const int side = anker_rc_comp(L, A.db, at, ts, t_len, q, &status);
					if(side < 0) { q.rc ^= 1; tmpl_out = -at; S0 = kma_score(L, A.db, at, ts, t_len, q, A.mq, -side, &status); }
					else if(side > 0) { tmpl_out = at; S0 = kma_score(L, A.db, at, ts, t_len, q, A.mq, side, &status); }
				}
			}
			if(status) atomicMax(&A.counters[1], 3ull);
		}
#ifdef KMAHIP_DIAG
		if(have) atomicMax(&A.counters[13], ((wall_clock64() - dbg_t0) << 32) | (unsigned long long) (task & 0xFFFFFFFFll));      // slowest phase A (10 ns ticks) and its task
#endif
		// ---- phase B: deferred wide DP problems, whole wave -----------------------------------------
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
#ifdef KMAHIP_DIAG
		const unsigned long long dbg_t1 = wall_clock64();
#endif
		// the deferred problems of this round: the LDS queues, then (long reads only) what did not fit them, spilled to this
		// wave's queues in HBM (written by single lanes, read by all: through L2)
		auto solve = [&](int *p0, int *p1, int *p2, int *p3, const int nq, const int nqn, const int nqt, const int nqx) {
			for(int e = 0; e < nqx; ++e) {
				// two columns per lane up to 128 columns (half the cells per step), four beyond
				const int *xe = p3 + 1 + e * QENT;
				const bool bnd = (xe[11] >> 1) != 0, wide2 = xe[6] - xe[5] <= 128;
				if(bnd) { if(wide2) nw_coop_x<2, true>(L, A.db, A, p3, e, tbuf); else nw_coop_x<4, true>(L, A.db, A, p3, e, tbuf); }
				else { if(wide2) nw_coop_x<2, false>(L, A.db, A, p3, e, tbuf); else nw_coop_x<4, false>(L, A.db, A, p3, e, tbuf); }
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
			for(int e = 0; e < nq; ++e) nw_coop<64>(L, A.db, A, p0, e, nq, tbuf);
			for(int e = 0; e < nqn; e += 4) nw_coop<16>(L, A.db, A, p1, e, nqn, tbuf);
			for(int e = 0; e < nqt; e += 8) nw_coop<8>(L, A.db, A, p2, e, nqt, tbuf);
		};
		auto collect = [&](const int *qu, const int cnt) {
			for(int e = 0; e < cnt; ++e) {
				const int *ent = qu + 1 + e * QENT;
				if(lane != ent[0]) continue;
				Aln &S = ent[1] ? S1 : S0;
				// a task that bailed out after queueing (gap too large, align.c:715) keeps its failure value
				if(S.len == 1 && S.match == 0 && S.score == 0) continue;
				S.score += ent[2]; S.len += ent[3]; S.match += ent[4]; S.tGaps += ent[5]; S.qGaps += ent[6];
				if(ent[11] & 1) S.pos -= ent[3] - ent[5];
			}
		};
		{
			const int nq = min(QCAP, queue[0]), nqn = min(QCAPN, queueN[0]), nqt = min(QCAPT, queueT[0]), nqx = min(QCAPX, queueX[0]);
			solve(queue, queueN, queueT, queueX, nq, nqn, nqt, nqx);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			collect(queue, nq); collect(queueN, nqn); collect(queueT, nqt); collect(queueX, nqx);
		}
		if(L.xq_cap) {
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
			const size_t cs = (size_t) L.xq_cap * QENT;
			int *p0 = L.xq - 1, *p1 = L.xq + cs - 1, *p2 = L.xq + 2 * cs - 1, *p3 = L.xq + 3 * cs - 1;
			const int nq = min(L.xq_cap, L.xq_cnt[0]), nqn = min(L.xq_cap, L.xq_cnt[1]), nqt = min(L.xq_cap, L.xq_cnt[2]), nqx = min(L.xq_cap, L.xq_cnt[3]);
			solve(p0, p1, p2, p3, nq, nqn, nqt, nqx);
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
			__builtin_amdgcn_wave_barrier();
			collect(p0, nq); collect(p1, nqn); collect(p2, nqt); collect(p3, nqx);
		}
#ifdef KMAHIP_DIAG
		if(lane == 0) atomicMax(&A.counters[14], ((wall_clock64() - dbg_t1) << 32) | (unsigned long long) (task & 0xFFFFFFFFll));         // slowest phase B
#endif
		// ---- phase C ---------------------------------------------------------------------------------
		if(!have || kind == 3) continue;
		task_finish<PEM>(A, task, kind, S0, S1, tmpl_out, t_len, qlen0, qlen1, k);
	}
}

// ---- the register-only kernel (round 4) ------------------------------------------------------------------------------------------
// align_tasks_kernel above is general -- it seeds what seed_tasks_kernel left out, settles strand ties, walks DP problems of any size
// in its lane -- and pays for that on every task: MEM arrays in HBM scratch (a dependent round trip per field), a lane struct and the
// arguments of kma_score in private memory (1 136 B of scratch per lane, 128 VGPRs, 4 waves / SIMD; counter traffic 11x the
// algorithmic bytes). But what nearly every task of a short-read sample needs is little: the <= SEEDS MEMs seed_tasks_kernel handed
// over, chainSeeds over those few, and tails / links that are provably diagonal (nw_diagonal) or one column wide. Stage 3a is
// therefore four launches:
//   align_fast_kernel    a lane per task, the MEMs, the chain and the running sums in registers -- no scratch, nothing of KMA_score
//                        in memory. A DP problem of 2..63 columns is not solved here: its descriptor goes to a DEVICE-WIDE queue
//                        of its width class (results are only ever summed into the task's figures, so the walk goes on), the task's
//                        sums so far to a row of `part`, the task to the pending list. Queue entries, pending and handed-on tasks
//                        are staged per wavefront in LDS and appended with one global atomic per list and flush (a lane's own
//                        atomic per entry on one address would cost more than the kernel).
//   dp_queue_kernel<W>   per width class, the cooperative anti-diagonal sweep nw_coop<W> over the whole queue -- dense: every
//                        segment of every wavefront has a problem -- adding each result into its task's row of `part`.
//   pend_finish_kernel   the filters of phase C for the pending tasks, from their rows.
//   align_tasks_kernel   the general kernel, over the list of tasks the first handed on: no MEMs from the seeding kernel (N-rich
//                        or repeat-rich reads, > SEEDS MEMs), a strand tie, a problem of 64 and more columns or a banded one, a
//                        chain that fails after queueing. It runs last and overwrites whatever the others left for such a task.
// Same arithmetic, statement for statement, as kma_score / chain_seeds above (KMA_score align.c:509-748, chainSeeds chain.c:79-260);
// tests/test_align_gpu.py runs both routes against the oracle (KMAHIP_ALIGN_FAST=0: the general kernel for every task).
constexpr int AC_DQ = 16, AC_PEND = 20, DQ_CLASSES = 4;      // counters: entries of the class queues (wide 33..63 columns, narrow 9..16, tiny 2..8, mid 17..32), pending tasks
constexpr int DQ_STAGE = 96, PEND_STAGE = 96, PUNT_STAGE = 64;      // per wavefront, in LDS
constexpr int PART_INTS = 16;                    // a pending task's row: S0 (6), S1 (6), kind, qlen0, qlen1, -

template <class T>
__device__ __forceinline__ T sel4(const T (&a)[SEEDS], int i) { return i == 0 ? a[0] : i == 1 ? a[1] : i == 2 ? a[2] : a[3]; }

struct FastCtx {
	const int *d;              // 25 ints in LDS
	int M, MM, U, W1;
	int *st_cnt;               // this wave's staging counters in LDS: [0] queue entries, [1] pending tasks, [2] handed-on tasks
	int *st_dq;                // DQ_STAGE entries of QENT ints
	int task32, at, mate, rc;  // the KMA_score call being run, for the queue entries
	int64_t rd;
	int n_queued;              // entries this call has staged
};

// nw_auto for a task of this kernel: the degenerate problems (nw.c:662-684), the provably diagonal ones and those of one query column
// at once; 2..63 columns to the staging queue of the wave (the result is zero for now); ok = false: the task is handed on
__device__ __forceinline__ Aln nw_auto_fast(FastCtx &C, const Lane &L, const uint64_t *ts, int t_len, const QView &q, int k,
                                            int t_s, int t_e, int q_s, int q_e, int tspan, int band, bool &ok) {
	const int ql = q_e - q_s;
	Aln r = {0, 0, 0, 0, 0, 0};
	ok = true;
	if(!(ql <= band || tspan <= band)) { ok = false; return r; }             // NW_band_score
	if(ql == 0 || tspan == 0) return nw_degenerate(tspan, ql, C.U, C.W1);
	if(nw_diagonal(L, ts, q, k, t_s, t_e, q_s, q_e, tspan, r)) return r;
	if(ql == 1 && tspan < 998) return nw_col1(L, ts, t_len, q, k, t_s, t_e, q_s);
	if(ql < WCOLS && tspan + ql < 1000) {
		const bool tiny = ql <= 8 && tspan < TBUF / 8;
		const bool narrow = !tiny && ql <= 16 && tspan < TBUF / 4;
		const bool mid = !tiny && !narrow && ql <= 32 && tspan < TBUF / 2;
		const int slot = atomicAdd(&C.st_cnt[0], 1);
		if(slot < DQ_STAGE) {
			int *e = C.st_dq + slot * QENT;
			e[0] = C.task32; e[1] = C.mate | ((tiny ? 2 : narrow ? 1 : mid ? 3 : 0) << 1); e[2] = k; e[3] = t_s; e[4] = t_e; e[5] = q_s; e[6] = q_e;
			e[7] = C.at; e[8] = (int) (C.rd & 0xFFFFFFFFll); e[9] = (int) (C.rd >> 32); e[10] = C.rc; e[11] = (k < 0) ? 1 : 0;
			++C.n_queued;
			return r;
		}
	}
	ok = false;
	return r;
}

// KMA_score on n <= SEEDS preseeded MEMs (1-based template start, query start, length), everything in registers.
// *punt: the task needs the general kernel.
__device__ __forceinline__ Aln kma_score_fast(FastCtx &C, const Lane &L, const uint64_t *ts, int t_len, const QView &q, int mq, int k,
                                              int n, const uint2 (&mem)[SEEDS], bool *punt) {
	const Aln FAIL = {0, 1, 0, 0, 0, 0};
	const int q_len = q.L, bw = 64;
	const int M = C.M, MM = C.MM, U = C.U, W1 = C.W1;
	C.n_queued = 0;
	int tS[SEEDS], tE[SEEDS], qS[SEEDS], qE[SEEDS], wt[SEEDS], sc[SEEDS], nx[SEEDS];
#pragma unroll
	for(int x = 0; x < SEEDS; ++x) {
		const int ln = (int) (mem[x].y >> 16);
		tS[x] = (int) mem[x].x; tE[x] = tS[x] + ln; qS[x] = (int) (mem[x].y & 0xFFFFu); qE[x] = qS[x] + ln; wt[x] = ln; sc[x] = 0; nx[x] = 0;
	}
	// chainSeeds (chain.c:79-260) over MEMs n-1 .. 0, the loops unrolled so that every array index is a constant
	int best = 0, second = 0, bestPos = n - 1;
#pragma unroll
	for(int i = SEEDS - 1; i >= 0; --i) {
		if(i < n) {
			const int wi = wt[i];
			const int weight = wi * M, tEnd = tE[i], qEnd = qE[i];
			int nxt = 0;
			int span = min(t_len - tEnd, q_len - qEnd);
			int gap = span - 1;
			gap = gap ? gap * U + W1 : W1;
			int sub = mism_score(span, k, M, MM);
			int score = weight + (sub < gap ? gap : sub);
#pragma unroll
			for(int j = i + 1; j < SEEDS; ++j) {
				if(j < n) {
					const int qSj = qS[j], tSj = tS[j];
					if(qEnd < qSj) {
						if(tEnd < tSj) {
							const int tGap = tSj - tEnd, qGap = qSj - qEnd;
							int g = abs(tGap - qGap);
							if(g) g = (g - 1) * U + W1;
							g += weight + sc[j] + mism_score(min(tGap, qGap), k, M, MM);
							if(score <= g) { score = g; nxt = j; }
						} else if(k <= tE[j] - tEnd) {
							int g = qSj - qEnd;
							if(g) g = (g - 1) * U + W1;
							g += weight + sc[j] - (tSj - tEnd) * M;
							if(score < g) { score = g; nxt = j; }
						}
					} else if(k <= qE[j] - qEnd) {
						const int tStart = tSj + qEnd - qSj;
						if(tEnd < tStart) {
							int g = tStart - tEnd;
							if(g) g = (g - 1) * U + W1;
							g += weight + sc[j] - (tStart - tEnd) * M;
							if(score < g) { score = g; nxt = j; }
						}
					}
				}
			}
			nx[i] = nxt;
			wt[i] = nxt ? (wi + sel4(wt, nxt) - k + 1) : (wi - (k - 1));
			sc[i] = score;
			span = min(tS[i], qS[i]);
			gap = span - 1;
			if(0 < gap) gap = gap * U + W1; else if(gap == 0) gap = W1; else gap = 0;
			sub = mism_score(span, k, M, MM);
			score += sub < gap ? gap : sub;
			if(best <= score) {
				if(nxt != bestPos) second = best;
				best = score; bestPos = i;
			} else if(second <= score && nxt != bestPos) {
				second = best;
			}
		}
	}
	unsigned mapQ = 0;
	if(0 < best) {
		const double wq = fmin(1.0, sel4(wt, bestPos) / 10.0);
		mapQ = (unsigned) ceil(40 * (1 - 1.0 * second / best) * wq * log((double) best));
	}
	if(mapQ < (unsigned) mq || best < k) return FAIL;

	// the chain from its best start: the MEM in hand in scalars (the reference's write-backs of a clipped start, align.c:694-711, are
	// only ever read for the MEM in hand). One loop for the leading tail (leadTailAln, align.c:53-131), every MEM with the link behind
	// it (:640-735) and the trailing tail (trailTailAln, :140-212), so that the DP dispatch exists once.
	int start = bestPos;
	int ctS = sel4(tS, start), ctE = sel4(tE, start), cqS = sel4(qS, start), cqE = sel4(qE, start);
	Aln S = {0, 0, 0, 0, 0, 0};
	S.pos = ctS - 1;
	for(int stage = 0;; stage = 1) {
		int kmode = 0, t_s = 0, t_e = 0, q_s = 0, q_e = 0, tspan = 0;
		bool dp = false, last = false;
		if(stage == 0) {
			t_e = ctS - 1; q_e = cqS;
			if(q_e) {
				if((q_e << 1) < t_e || (q_e + bw) < t_e) t_s = t_e - (q_e + (q_e < bw ? q_e : bw));
				else if((t_e << 1) < q_e || (t_e + bw) < q_e) q_s = q_e - (t_e + (t_e < bw ? t_e : bw));
				if(t_e - t_s > 0 && q_e - q_s > 0) { dp = true; kmode = -1 - (t_s == 0); tspan = t_e - t_s; }
			}
		} else {
			S.len += cqE - cqS; S.match += cqE - cqS;
			S.score += (cqE - cqS) * C.d[0];              // (a MEM holds no N and the matrix is uniform on its diagonal: the host checks)
			const int nxt = sel4(nx, start);
			q_s = cqE; t_s = ctE - 1;
			if(nxt) {
				start = nxt;
				int qSn = sel4(qS, start), tSn = sel4(tS, start);
				const int tEn = sel4(tE, start);
				if(qSn < q_s) { tSn += q_s - qSn; qSn = q_s; }
				int t_l;
				t_e = tSn - 1;
				if(t_e < t_s) {
					if(t_s <= tEn) { qSn += t_s - t_e; t_e = t_s; t_l = 0; }
					else t_l = t_len - t_s + t_e;
				} else t_l = t_e - t_s;
				ctS = tSn; ctE = tEn; cqS = qSn; cqE = sel4(qE, start);
				q_e = qSn;
				if(abs(t_l - q_e + q_s) * U > q_len * M || t_l > q_len || q_e - q_s > (q_len >> 1)) {
					// (align.c:715; the general kernel drops what such a call has queued: it takes the task)
					if(C.n_queued) *punt = true;
					return FAIL;
				}
				if(t_l > 0 || q_e - q_s > 0) { dp = true; kmode = 0; tspan = t_l; }
			} else {
				last = true;
				q_e = q_len; t_e = t_len;
				if(((q_len - q_s) << 1) < (t_len - t_s) || (q_len - q_s + bw) < (t_len - t_s)) {
					t_e = q_len - q_s; t_e = t_s + (t_e + (t_e < bw ? t_e : bw));
				} else if(((t_len - t_s) << 1) < (q_len - q_s) || (t_len - t_s + bw) < (q_len - q_s)) {
					q_e = t_len - t_s; q_e = q_s + (q_e + (q_e < bw ? q_e : bw));
				}
				if(t_e - t_s > 0 && q_e - q_s > 0) { dp = true; kmode = 1 + (t_e == t_len); tspan = t_e - t_s; }
			}
		}
		if(dp) {
			const int band = abs(tspan - q_e + q_s) + bw;
			bool ok;
			const Aln r = nw_auto_fast(C, L, ts, t_len, q, kmode, t_s, t_e, q_s, q_e, tspan, band, ok);
			if(!ok) { *punt = true; return FAIL; }
			if(stage == 0) S.pos -= r.len - r.tGaps;
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
		if(last) break;
	}
	return S;
}

constexpr int FTHREADS = 256;
#ifndef KMAHIP_FAST_WAVES
#define KMAHIP_FAST_WAVES 4
#endif

// the staged entries of a wavefront to the device-wide lists: one global atomic per list that has something. Called by all 64 lanes.
__device__ void fast_flush(const AlignArgs &A, int *st_cnt, const int *st_dq, const int *st_pend, const int *st_punt) {
	const int lane = (int) (threadIdx.x & 63);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const int nq = min(DQ_STAGE, st_cnt[0]), np = min(PEND_STAGE, st_cnt[1]), nu = min(PUNT_STAGE, st_cnt[2]);
	for(int c0 = 0; c0 < nq; c0 += 64) {
		const int i = c0 + lane;
		const int cls = i < nq ? (st_dq[i * QENT + 1] >> 1) : -1;
		for(int c = 0; c < DQ_CLASSES; ++c) {
			const unsigned long long m = __ballot(cls == c);
			if(!m) continue;
			const int leader = __ffsll((long long) m) - 1;
			unsigned long long first = 0;
			if(lane == leader) first = atomicAdd(&A.counters[AC_DQ + c], (unsigned long long) __popcll(m));
			first = __shfl(first, leader);
			if(cls == c) {
				const int64_t at = (int64_t) first + __popcll(m & ((1ull << lane) - 1ull));
				if(at < A.dq_cap) {
					int4 *dst = (int4 *) (A.dq + ((size_t) c * A.dq_cap + at) * QENT);
					const int4 *src = (const int4 *) (st_dq + i * QENT);
					dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
				} else {
					// no room (the host sized the queues for the usual sample): the general kernel takes the task, and its row -- written
					// by this wavefront before it staged the entry -- says so to pend_finish_kernel
					A.slow_list[atomicAdd(&A.counters[2], 1ull)] = st_dq[i * QENT];
					A.part[(int64_t) st_dq[i * QENT] * PART_INTS + 15] = 1;
				}
			}
		}
	}
	for(int c0 = 0; c0 < np; c0 += 64) {
		const int i = c0 + lane;
		const unsigned long long m = __ballot(i < np);
		unsigned long long first = 0;
		if(lane == 0) first = atomicAdd(&A.counters[AC_PEND], (unsigned long long) __popcll(m));
		first = __shfl(first, 0);
		if(i < np) A.pend[first + lane] = st_pend[i];
	}
	for(int c0 = 0; c0 < nu; c0 += 64) {
		const int i = c0 + lane;
		const unsigned long long m = __ballot(i < nu);
		unsigned long long first = 0;
		if(lane == 0) first = atomicAdd(&A.counters[2], (unsigned long long) __popcll(m));
		first = __shfl(first, 0);
		if(i < nu) A.slow_list[first + lane] = st_punt[i];
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	if(lane < 3) st_cnt[lane] = 0;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

template <bool PEM>
__global__ __launch_bounds__(FTHREADS, KMAHIP_FAST_WAVES) void align_fast_kernel(const AlignArgs A) {
	__shared__ int s_d[25];
	__shared__ int s_cnt[(FTHREADS / 64) * 4];
	__shared__ __attribute__((aligned(16))) int s_dq[(FTHREADS / 64) * DQ_STAGE * QENT];
	__shared__ int s_pend[(FTHREADS / 64) * PEND_STAGE];
	__shared__ int s_punt[(FTHREADS / 64) * PUNT_STAGE];
	if(threadIdx.x < 25) s_d[threadIdx.x] = A.d[threadIdx.x];
	if(threadIdx.x < (FTHREADS / 64) * 4) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const int wave = (int) (threadIdx.x >> 6), lane = (int) (threadIdx.x & 63);
	int *const st_cnt = s_cnt + wave * 4, *const st_dq = s_dq + wave * DQ_STAGE * QENT, *const st_pend = s_pend + wave * PEND_STAGE, *const st_punt = s_punt + wave * PUNT_STAGE;
	FastCtx C;
	C.d = s_d; C.M = A.M; C.MM = A.MM; C.U = A.U; C.W1 = A.W1; C.st_cnt = st_cnt; C.st_dq = st_dq;
	C.task32 = 0; C.at = 0; C.mate = 0; C.rc = 0; C.rd = 0; C.n_queued = 0;
	// (what the shared helpers -- nw_diagonal, nw_col1 -- read of a Lane; both are inlined here, the struct never exists in memory)
	Lane L;
	L.s32 = nullptr; L.s64 = nullptr; L.r32 = nullptr; L.r64 = nullptr; L.lanes = 0; L.cap1 = 0; L.ncols = 0;
	L.d = s_d; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1; L.wide = nullptr; L.queue = nullptr; L.xq = nullptr; L.xq_cnt = nullptr; L.xq_cap = 0;
	L.q_at = 0; L.q_mate = 0; L.q_rd = 0; L.ablate = 0; L.gap_m_max = A.gap_m_max; L.diag_uniform = 1; L.cnt = nullptr;
	const int64_t n_tasks = A.T_off[A.n_reads];
	if(n_tasks > A.tasks_cap) return;
	const int k = (int) A.db.kmersize;
	const int slots = PEM ? 2 : 1;
	const int64_t stride = (int64_t) gridDim.x * FTHREADS;
	// (the trip count is the same for the lanes of a wavefront: the flush inside is a wave-wide step)
	for(int64_t base = (int64_t) blockIdx.x * FTHREADS + (threadIdx.x & ~63u); base < n_tasks; base += stride) {
		const int64_t task = base + lane;
		if(task < n_tasks) {
			int kind = 0;                  // 0 nothing, 1 single record, 2 couple, 3 left to the long-read pipeline
			bool punt = false;
			int queued = 0;
			Aln S0 = {0, 0, 0, 0, 0, 0}, S1 = {0, 1, 0, 0, 0, 0};
			int t_len = 0, qlen0 = 0, qlen1 = 0;
			const int64_t r = A.t_rec[task];
			const int tmpl_out = A.T[task];
			const int at = abs(tmpl_out);
			const int rcf = A.rc_flag[r];
			bool couple = false;
			int64_t rd = r;
			int orient;
			C.task32 = (int) task; C.at = at;
			if(PEM) {
				const int64_t p0 = r & ~1ll;
				const int mate_r = A.rec_mate[r];
				rd = p0 + max(0, mate_r);
				orient = A.rec_rc[r];
				couple = (r & 1) && A.rec_mate[p0] >= 0 && mate_r >= 0 && A.T_off[p0 + 1] == A.T_off[p0];
				if(!couple && !(rcf != 0 && mate_r >= 0)) rd = -1;
			} else {
				orient = (A.flag[r] & 16) ? 1 : 0;
				if(long_routed(A, r)) { kind = 3; rd = -1; }
				else if(rcf == 0) rd = -1;
			}
			if(rd >= 0 || couple) {
				t_len = A.db.tlen[at];
				const uint64_t *ts = A.db.tseq + A.db.tseq_off[at];
				if(PEM && couple) {
					// alnFragsPenaltyPE, alnfrags.c:1630-1775: both records of the pair against this candidate, both flipped once the
					// list has reached its first negative id (:1633-1647)
					kind = 2;
					int rcstate = 0;
					for(int64_t j = A.T_off[r]; j <= task; ++j) if(A.T[j] < 0) { rcstate = 1; break; }
					for(int m = 0; m < 2 && !punt; ++m) {
						const int64_t rec = (r & ~1ll) + m;
						const int64_t rdm = (r & ~1ll) + A.rec_mate[rec];
						QView q;
						q.w = A.seq + A.seq_off[rdm]; q.L = A.len[rdm]; q.rc = A.rec_rc[rec] ^ rcstate;
						q.N = A.N + A.N_off[rdm]; q.nN = (int) (A.N_off[rdm + 1] - A.N_off[rdm]);
						Aln st = {0, 1, 0, 0, 0, 0};
						if(q.L >= k) {
							const int pre = A.seed_n[task * 2 + m];
							if(pre < 0) punt = true;
							else if(pre > 0) {
								const uint4 *mp = (const uint4 *) (A.seed_mem + (task * 2 + m) * SEEDS);
								const uint4 m01 = mp[0], m23 = mp[1];
								const uint2 mem[SEEDS] = {make_uint2(m01.x, m01.y), make_uint2(m01.z, m01.w), make_uint2(m23.x, m23.y), make_uint2(m23.z, m23.w)};
								C.mate = m; C.rc = q.rc; C.rd = rdm;
								st = kma_score_fast(C, L, ts, t_len, q, A.mq, k, pre, mem, &punt);
								queued += C.n_queued;
							}
						}
						if(m == 0) { S0 = st; qlen0 = q.L; } else { S1 = st; qlen1 = q.L; }
					}
				} else if(A.len[rd] >= k) {
					kind = 1;
					if(rcf < 0) punt = true;          // strand tie: anker_rc_comp
					else {
						QView q;
						q.w = A.seq + A.seq_off[rd]; q.L = A.len[rd]; q.rc = orient;
						q_set_bounds(q, A.q_start, A.q_end, rd);
						q.N = A.N + A.N_off[rd]; q.nN = (int) (A.N_off[rd + 1] - A.N_off[rd]);
						qlen0 = q.L;
						const int pre = A.seed_n[task * slots];
						if(pre < 0) punt = true;
						else if(pre == 0) S0 = Aln{0, 1, 0, 0, 0, 0};
						else {
							const uint4 *mp = (const uint4 *) (A.seed_mem + task * slots * SEEDS);
							const uint4 m01 = mp[0], m23 = mp[1];
							const uint2 mem[SEEDS] = {make_uint2(m01.x, m01.y), make_uint2(m01.z, m01.w), make_uint2(m23.x, m23.y), make_uint2(m23.z, m23.w)};
							C.mate = 0; C.rc = q.rc; C.rd = rd;
							S0 = kma_score_fast(C, L, ts, t_len, q, A.mq, k, pre, mem, &punt);
							queued = C.n_queued;
						}
					}
				}
			}
			if(kind != 3) {
				if(punt) {
					const int slot = atomicAdd(&st_cnt[2], 1);
					if(slot < PUNT_STAGE) st_punt[slot] = (int) task;
					else A.slow_list[atomicAdd(&A.counters[2], 1ull)] = task;
				} else if(queued) {
					// the sums so far; the queues' results are added to this row, pend_finish_kernel filters it
					int4 *row = (int4 *) (A.part + task * PART_INTS);
					row[0] = make_int4(S0.score, S0.len, S0.pos, S0.match);
					row[1] = make_int4(S0.tGaps, S0.qGaps, S1.score, S1.len);
					row[2] = make_int4(S1.pos, S1.match, S1.tGaps, S1.qGaps);
					row[3] = make_int4(kind, qlen0, qlen1, 0);
					const int slot = atomicAdd(&st_cnt[1], 1);
					if(slot < PEND_STAGE) st_pend[slot] = (int) task;
					else A.pend[atomicAdd(&A.counters[AC_PEND], 1ull)] = (int) task;
				} else task_finish<PEM>(A, task, kind, S0, S1, tmpl_out, t_len, qlen0, qlen1, k);
			}
		}
		// a round can stage 64 x 5 entries at most, usually one or two: flush when half full (a lane that finds no slot hands its task on)
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		if(st_cnt[0] >= DQ_STAGE / 2 || st_cnt[1] >= PEND_STAGE / 2 || st_cnt[2] >= PUNT_STAGE / 2 || base + stride >= n_tasks) fast_flush(A, st_cnt, st_dq, st_pend, st_punt);
	}
}

// the queue of one width class: nw_coop<W> over every entry, the result added to the task's row
template <int W>
__global__ __launch_bounds__(256) void dp_queue_kernel(const AlignArgs A, int cls) {
	__shared__ int s_d[25];
	__shared__ uint8_t s_tbuf[4 * TBUF];
	if(threadIdx.x < 25) s_d[threadIdx.x] = A.d[threadIdx.x];
	__syncthreads();
	constexpr int G = 64 / W;
	const int wave = (int) (threadIdx.x >> 6), lane = (int) (threadIdx.x & 63);
	Lane L;
	L.s32 = nullptr; L.s64 = nullptr; L.r32 = nullptr; L.r64 = nullptr; L.lanes = 0; L.cap1 = 0; L.ncols = 0;
	L.d = s_d; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1; L.wide = nullptr; L.queue = nullptr; L.xq = nullptr; L.xq_cnt = nullptr; L.xq_cap = 0;
	L.q_at = 0; L.q_mate = 0; L.q_rd = 0; L.ablate = 0; L.gap_m_max = A.gap_m_max; L.diag_uniform = 1; L.cnt = nullptr;
	const int count = (int) min((unsigned long long) A.dq_cap, A.counters[AC_DQ + cls]);
	int *const qu = A.dq + (size_t) cls * A.dq_cap * QENT - 1;          // (nw_coop counts its entries from qu + 1)
	uint8_t *const tbuf = s_tbuf + wave * TBUF;
	const int n_waves = (int) gridDim.x * 4;
	for(int first = ((int) blockIdx.x * 4 + wave) * G; first < count; first += n_waves * G) {
		nw_coop<W>(L, A.db, A, qu, first, count, tbuf);
		const int g = lane / W;
		if((lane & (W - 1)) == 0 && first + g < count) {
			const int *e = qu + 1 + (first + g) * QENT;
			int *row = A.part + (int64_t) e[0] * PART_INTS + (e[1] & 1) * 6;
			atomicAdd(&row[0], e[2]); atomicAdd(&row[1], e[3]); atomicAdd(&row[3], e[4]); atomicAdd(&row[4], e[5]); atomicAdd(&row[5], e[6]);
			if(e[11] & 1) atomicAdd(&row[2], -(e[3] - e[5]));
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();          // (the staged template bases are the next problem's)
	}
}

template <bool PEM>
__global__ __launch_bounds__(256) void pend_finish_kernel(const AlignArgs A) {
	const int64_t n = (int64_t) A.counters[AC_PEND];
	const int k = (int) A.db.kmersize;
	for(int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t) gridDim.x * 256) {
		const int64_t task = A.pend[i];
		const int4 *row = (const int4 *) (A.part + task * PART_INTS);
		const int4 a = row[0], b = row[1], c = row[2], d = row[3];
		if(d.w) continue;          // (handed on after all: a queue had no room)
		const Aln S0 = {a.x, a.y, a.z, a.w, b.x, b.y}, S1 = {b.z, b.w, c.x, c.y, c.z, c.w};
		const int tmpl_out = A.T[task];
		task_finish<PEM>(A, task, d.x, S0, S1, tmpl_out, A.db.tlen[abs(tmpl_out)], d.y, d.z, k);
	}
}

// task -> owning record, so that the task kernel needs one coalesced load instead of a 23-step binary search
__global__ __launch_bounds__(256) void task_map_kernel(const int64_t *T_off, int64_t n, int32_t *t_rec, int64_t tasks_cap, unsigned long long *counters) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= n) return;
	if(T_off[n] > tasks_cap) { if(r == 0) atomicMax(&counters[1], 2ull); return; }
	const int64_t e = T_off[r + 1];
	for(int64_t t = T_off[r]; t < e; ++t) t_rec[t] = (int32_t) r;
}

struct ReduceArgs {
	int64_t n_reads;
	const int32_t *rc_flag, *flag;
	const int64_t *T_off;
	const int32_t *T;
	const int32_t *t_score, *t_alen, *t_start, *t_end, *t_tmpl;
	const double *t_norm;
	int k;
	double scoreT;
	int32_t *n_hits, *best_score, *out_flag, *h_tmpl, *h_score, *h_start, *h_end;
	unsigned long long *alignment_scores, *uniq_alignment_scores;
	// private copies of the two vectors (copy c at priv + c * 2 * DB_size): every read adds into one of them, chosen by its
	// workgroup -- 11 M u64 atomics on a few thousand addresses otherwise queue up on single L2 lines
	unsigned long long *priv;
	int priv_copies;
	int64_t DB_size;
	// paired-end mode
	const int32_t *rec_mate, *rec_rc;
	int32_t *out_rc;      // may be NULL. bit 0: the fragment filed for this record is the reverse complement of the original read;
	                      // bit 1 (proper pair only): the pair's record is written second slot first (alnfrags.c:1807-1812)
	int pe_mode, PE, apm;
	int64_t tasks_cap;    // as AlignArgs.tasks_cap
	int32_t *pe_kind;     // per pair: 0 none / records handled singly, 1 proper pair, 2 unmated, 3 first only, 4 second only
	int32_t *t_score_w, *t_alen_w, *t_start_w, *t_end_w, *t_tmpl_w;   // writable views (unmated shuffle)
};

// per pair: the tail of alnFragsPenaltyPE (alnfrags.c:1777-1970) + update_Scores_pe / update_Scores_se with
// minFrac == 1. Hits of the pair are written into the second record's slice [T_off[p0+1], ...): proper pair:
// n shared hits; unmated: the first record's hits followed by the second's; single: that record's hits.
__device__ void reduce_couple(const ReduceArgs &R, int64_t p0) {
	const int64_t o = R.T_off[p0 + 1], nT = R.T_off[p0 + 2] - o;
	int best = 0, best_r = 0, comp = 0, rcstate = 0;
	for(int64_t i = 0; i < nT; ++i) {
		best = max(best, R.t_score[o + i]); best_r = max(best_r, R.t_alen[o + i]);
		comp = max(comp, R.t_tmpl[o + i] + R.t_score[o + i]);
		if(R.T[o + i] < 0) rcstate = 1;
	}
	int fA = R.flag[p0], fB = R.flag[p0 + 1], kind = 0, nA = 0, nB = 0, sA = 0, sB = 0;
	// Orientation of the two fragments as they end up in the frag record, relative to the S2 records: a negative candidate
	// makes alnFragsPenaltyPE reverse-complement both mates (alnfrags.c:1629-1643); a mate whose first kept template is
	// positive is turned back (with the flag toggled), one whose first kept template is negative stays turned.
	int relA = 0, relB = 0, swap_rec = 0;
	// -apm u (alnFragsUnionPE, alnfrags.c:1220-1594): a proper pair where a template holds BOTH mates' best scores (:1410-1446; its
	// score the sum of the two), else the unmated / one-mate branches below, which the two functions share
	int u_hits = 0;
	if(R.apm == 1 && best && best_r) for(int64_t i = 0; i < nT; ++i) u_hits += (best <= R.t_score[o + i] && best_r <= R.t_alen[o + i]);
	if(best || best_r) {
		if(R.apm == 1 ? u_hits != 0 : (comp && 1.0 * (best + best_r) <= comp + R.PE)) {
			const int pe = R.apm == 1 ? 0 : R.PE;
			const int bestScore = R.apm == 1 ? best + best_r : comp + R.PE;
			int first = 0, h = 0, c = 0;
			for(int64_t i = 0; i < nT; ++i) {
				const bool in = R.apm == 1 ? (best <= R.t_score[o + i] && best_r <= R.t_alen[o + i]) : (R.t_score[o + i] && R.t_alen[o + i]);
				if(in) { if(!h) first = R.T[o + i]; ++h; }
			}
			const bool swapped = h && first < 0;
			if(!swapped && rcstate) { fA ^= 48; fB ^= 48; }
			if(swapped) { relA = relB = 1; swap_rec = 2; }
			for(int64_t i = 0; i < nT; ++i) {
				const int bt = R.t_score[o + i], btr = R.t_alen[o + i];
				if(bt && btr && bt + btr + pe == bestScore && (R.apm != 1 || (best <= bt && best_r <= btr))) {
					const int tm = swapped ? -R.T[o + i] : R.T[o + i];
					R.h_tmpl[o + c] = tm; R.h_score[o + c] = bestScore; R.h_start[o + c] = R.t_start[o + i]; R.h_end[o + c] = R.t_end[o + i];
					if(R.alignment_scores) atomicAdd(&R.alignment_scores[abs(tm)], (unsigned long long) bestScore);
					++c;
				}
			}
			if(c == 1 && R.uniq_alignment_scores) atomicAdd(&R.uniq_alignment_scores[abs(R.h_tmpl[o])], (unsigned long long) bestScore);
			kind = 1; nA = nB = c; sA = sB = bestScore;
		} else if(best && best_r) {
			// unmated pair, alnfrags.c:1820-1891, restated literally (incl. that the first record's score array is
			// not part of the exchange and the `+ end` pointer shift). Inputs are 1-based views X1(i) = X[o + i - 1]
			// of the task arrays; outputs go to the hit arrays, which the reference overlays on the same memory
			// without ever reading a slot it has overwritten.
			int32_t *mt = R.t_tmpl_w + o, *b2 = R.t_alen_w + o, *s1 = R.t_start_w + o, *e1 = R.t_end_w + o;
			const int32_t *b1 = R.t_score + o;
			for(int64_t i = 0; i < nT; ++i) mt[i] = R.T[o + i];
			int h = 0, hr = 0, ti = 1, endp = (int) nT;
			while(ti <= endp) {
				if(best <= b1[ti - 1]) {
					R.h_tmpl[o + h] = mt[ti - 1]; R.h_score[o + h] = b1[ti - 1]; R.h_start[o + h] = s1[ti - 1]; R.h_end[o + h] = e1[ti - 1];
					++h; ++ti;
				} else if(best_r <= b2[ti - 1]) {
					int x;
					x = mt[ti - 1]; mt[ti - 1] = mt[endp - 1]; mt[endp - 1] = x;
					x = b2[ti - 1]; b2[ti - 1] = b2[endp - 1]; b2[endp - 1] = x;
					x = s1[ti - 1]; s1[ti - 1] = s1[endp - 1]; s1[endp - 1] = x;
					x = e1[ti - 1]; e1[ti - 1] = e1[endp - 1]; e1[endp - 1] = x;
					++hr; --endp;
				} else ++ti;
			}
			if(rcstate) { fA ^= 48; fB ^= 48; }     // ^16 ^32 on both (:1863-1876; the sign tests there look at scores)
			if(fA & 2) { fA ^= 2; fB ^= 2; }
			int c = 0, c2 = 0;
			for(int i = 0; i < h; ++i) if(R.h_score[o + i] == best) {
				R.h_tmpl[o + c] = R.h_tmpl[o + i]; R.h_start[o + c] = R.h_start[o + i]; R.h_end[o + c] = R.h_end[o + i]; R.h_score[o + c] = best;
				if(R.alignment_scores) atomicAdd(&R.alignment_scores[abs(R.h_tmpl[o + c])], (unsigned long long) best);
				++c;
			}
			if(c == 1 && R.uniq_alignment_scores) atomicAdd(&R.uniq_alignment_scores[abs(R.h_tmpl[o])], (unsigned long long) best);
			for(int i = 0; i < hr; ++i) {
				const int pos = endp + i;            // 1-based; position 0 is the reference's unused slot
				if(pos < 1 || pos > nT || b2[pos - 1] != best_r) continue;
				R.h_tmpl[o + c + c2] = mt[pos - 1]; R.h_start[o + c + c2] = s1[pos - 1]; R.h_end[o + c + c2] = e1[pos - 1]; R.h_score[o + c + c2] = best_r;
				if(R.alignment_scores) atomicAdd(&R.alignment_scores[abs(mt[pos - 1])], (unsigned long long) best_r);
				++c2;
			}
			if(c2 == 1 && R.uniq_alignment_scores) atomicAdd(&R.uniq_alignment_scores[abs(R.h_tmpl[o + c])], (unsigned long long) best_r);
			kind = 2; nA = c; nB = c2; sA = best; sB = best_r;
		} else {
			const bool first = best != 0;
			const int bscore = first ? best : best_r;
			int h = 0, c = 0, t0 = 0;
			for(int64_t i = 0; i < nT; ++i) { const int sc = first ? R.t_score[o + i] : R.t_alen[o + i]; if(sc) { if(!h) t0 = R.T[o + i]; ++h; } }
			bool neg = false;
			if(first) {
				if(h && t0 < 0) { neg = true; relA = 1; } else if(rcstate) { fA ^= 16; fB ^= 32; }
				fA |= 8; fB ^= 4;
				if(fA & 2) { fA ^= 2; fB ^= 2; }
			} else {
				if(rcstate) { fA ^= 32; fB ^= 16; }
				fB |= 8; fA ^= 4;
				if(fB & 2) { fA ^= 2; fB ^= 2; }
			}
			for(int64_t i = 0; i < nT; ++i) {
				const int sc = first ? R.t_score[o + i] : R.t_alen[o + i];
				if(sc && sc == bscore) {
					const int tm = neg ? -R.T[o + i] : R.T[o + i];
					R.h_tmpl[o + c] = tm; R.h_score[o + c] = bscore; R.h_start[o + c] = R.t_start[o + i]; R.h_end[o + c] = R.t_end[o + i];
					if(R.alignment_scores) atomicAdd(&R.alignment_scores[abs(tm)], (unsigned long long) bscore);
					++c;
				}
			}
			if(c == 1 && R.uniq_alignment_scores) atomicAdd(&R.uniq_alignment_scores[abs(R.h_tmpl[o])], (unsigned long long) bscore);
			kind = first ? 3 : 4;
			if(first) { nA = c; sA = best; } else { nB = c; sB = best_r; }
		}
	}
	R.pe_kind[p0 >> 1] = kind;
	R.n_hits[p0] = nA; R.n_hits[p0 + 1] = nB; R.best_score[p0] = sA; R.best_score[p0 + 1] = sB;
	R.out_flag[p0] = fA; R.out_flag[p0 + 1] = fB;
	if(R.out_rc) {
		R.out_rc[p0] = ((R.rec_rc[p0] != 0) != (relA != 0) ? 1 : 0) | swap_rec;
		R.out_rc[p0 + 1] = ((R.rec_rc[p0 + 1] != 0) != (relB != 0) ? 1 : 0) | swap_rec;
	}
}

// per read: hit filter of alnFragsSE (alnfrags.c:1165-1215) + update_Scores with
// minFrac == 1.0 (updatescores.c:217-234, :275-277)
__global__ __launch_bounds__(256) void fold_scores_kernel(const unsigned long long *priv, int copies, int64_t D,
                                                            unsigned long long *alignment_scores, unsigned long long *uniq_alignment_scores) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= 2 * D) return;
	unsigned long long sum = 0;
	for(int c = 0; c < copies; ++c) sum += priv[(int64_t) c * 2 * D + i];
	if(!sum) return;
	if(i < D) { if(alignment_scores) alignment_scores[i] += sum; }
	else if(uniq_alignment_scores) uniq_alignment_scores[i - D] += sum;
}

// one single-end record (alnFragsSE's tail + update_Scores with minFrac == 1, updatescores.c:203-298): keep the hits with the
// best normalised score or the best read score, add their scores to the ConClave vectors through add_as / add_us
template <class FA, class FU>
__device__ __forceinline__ void reduce_single(const ReduceArgs &R, int64_t r, int64_t o, int64_t e, int fl, FA add_as, FU add_us) {
	int nh = 0, bestRead = 0;
	if(e > o) {
		double bestScore = 0.0;
		for(int64_t t = o; t < e; ++t) {
			const int rs = R.t_score[t];
			const double sc = R.t_norm[t];
			if(R.k < rs && R.scoreT <= sc) {
				if(bestScore < sc) bestScore = sc;
				if(bestRead < rs) bestRead = rs;
			}
		}
		if(bestRead > R.k) {
			for(int64_t t = o; t < e; ++t) {
				const int rs = R.t_score[t];
				if(!(R.k < rs && R.scoreT <= R.t_norm[t])) continue;
				const double ms = (double) (rs / R.t_alen[t]);
				if(ms == bestScore || rs == bestRead) {
					const int64_t w = o + nh;
					const int tm = R.t_tmpl[t];
					R.h_tmpl[w] = tm; R.h_score[w] = rs; R.h_start[w] = R.t_start[t]; R.h_end[w] = R.t_end[t];
					add_as(abs(tm), rs);
					++nh;
				}
			}
			if(nh == 1) add_us(abs(R.h_tmpl[o]), bestRead);
		} else {
			fl |= 4;
			bestRead = 0;
		}
	}
	R.n_hits[r] = nh; R.best_score[r] = (nh > 0) ? bestRead : 0; R.out_flag[r] = fl;
	// single records are filed in the orientation stage 2 passed on (paired: rec_rc; single end: flag & 16)
	if(R.out_rc) R.out_rc[r] = R.pe_mode ? (R.rec_rc[r] != 0) : ((R.flag[r] & 16) != 0);
}

__global__ __launch_bounds__(256) void reduce_reads_kernel(const ReduceArgs R0) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= R0.n_reads) return;
	if(R0.T_off[R0.n_reads] > R0.tasks_cap) {
		// candidate list beyond the capacity (status 2 is set): no hits, so that whatever runs next on the stream stays in bounds
		R0.n_hits[r] = 0; R0.best_score[r] = 0; R0.out_flag[r] = R0.flag[r];
		if(R0.out_rc) R0.out_rc[r] = 0;
		if(R0.pe_mode && !(r & 1)) R0.pe_kind[r >> 1] = 0;
		return;
	}
	ReduceArgs R = R0;
	if(R.priv_copies) {
		unsigned long long *base = R.priv + (int64_t) (blockIdx.x % (unsigned) R.priv_copies) * 2 * R.DB_size;
		if(R.alignment_scores) R.alignment_scores = base;
		if(R.uniq_alignment_scores) R.uniq_alignment_scores = base + R.DB_size;
	}
	const int64_t o = R.T_off[r], e = R.T_off[r + 1];
	const int fl = R.flag[r];
	if(R.pe_mode) {
		const int64_t p0 = r & ~1ll;
		const bool couple = R.rec_mate[p0] >= 0 && R.rec_mate[p0 + 1] >= 0 && R.T_off[p0 + 1] == R.T_off[p0];
		if(couple) {
			if(r & 1) reduce_couple(R, p0);
			return;
		}
		if(!(r & 1)) R.pe_kind[r >> 1] = 0;
		if(R.rec_mate[r] < 0) { R.n_hits[r] = 0; R.best_score[r] = 0; R.out_flag[r] = fl; if(R.out_rc) R.out_rc[r] = 0; return; }
	}
	reduce_single(R, r, o, e, fl, [&](int t, int v) { if(R.alignment_scores) atomicAdd(&R.alignment_scores[t], (unsigned long long) v); },
	              [&](int t, int v) { if(R.uniq_alignment_scores) atomicAdd(&R.uniq_alignment_scores[t], (unsigned long long) v); });
}

// single-end batches against a database small enough for LDS (two u32 vectors of DB_size entries): every workgroup sums the
// scores of its reads in LDS and flushes what is non-zero once -- the one-thread-per-read form above issues one or two u64
// atomics per read and is bound by them (21 M for 10 M reads)
constexpr int RL_THREADS = 1024;
constexpr int RL_MAX_DB = 6144;
__global__ __launch_bounds__(RL_THREADS) void reduce_reads_lds_kernel(const ReduceArgs R) {
	__shared__ uint32_t s_acc[2 * RL_MAX_DB];
	const int D = (int) R.DB_size;
	if(R.T_off[R.n_reads] > R.tasks_cap) {
		for(int64_t r = (int64_t) blockIdx.x * RL_THREADS + threadIdx.x; r < R.n_reads; r += (int64_t) gridDim.x * RL_THREADS) {
			R.n_hits[r] = 0; R.best_score[r] = 0; R.out_flag[r] = R.flag[r];
			if(R.out_rc) R.out_rc[r] = 0;
		}
		return;
	}
	for(int i = threadIdx.x; i < 2 * D; i += RL_THREADS) s_acc[i] = 0;
	__syncthreads();
	for(int64_t r = (int64_t) blockIdx.x * RL_THREADS + threadIdx.x; r < R.n_reads; r += (int64_t) gridDim.x * RL_THREADS) {
		reduce_single(R, r, R.T_off[r], R.T_off[r + 1], R.flag[r], [&](int t, int v) { atomicAdd(&s_acc[t], (uint32_t) v); },
		              [&](int t, int v) { atomicAdd(&s_acc[D + t], (uint32_t) v); });
	}
	__syncthreads();
	for(int i = threadIdx.x; i < 2 * D; i += RL_THREADS) {
		const uint32_t v = s_acc[i];
		if(!v) continue;
		if(i < D) { if(R.alignment_scores) atomicAdd(&R.alignment_scores[i], (unsigned long long) v); }
		else if(R.uniq_alignment_scores) atomicAdd(&R.uniq_alignment_scores[i - D], (unsigned long long) v);
	}
}

// ---- stage 3c, per read: KMA() with traceback (align.c:214-507; NW nw.c:26-309, NW_band :310-640) ----------------
// One lane per read that ConClave filed under a template. Same seeds / chain / joins as kma_score above, but the move
// matrix E is kept (bytes, per-lane HBM scratch) and walked into alignment columns, which leave the kernel as a
// run-length list of (=, X, I, D) -- the classes of makeCigar (sam.c:57-78) and everything alnToMat needs besides
// the read itself (assembly.c:1317-1444).
struct TraceArgs {
	DevDB db;
	int64_t n_reads;
	const uint64_t *seq;
	const int64_t *seq_off;
	const int32_t *len;
	const int32_t *N;
	const int64_t *N_off;
	const int32_t *flag;         // kmahip_hits.rc: bit 0 = the filed fragment is the reverse complement of the read
	const int32_t *tmpl;         // ConClave's signed template per read (0: none)
	const uint8_t *tmpl_ok;      // per template: assemble it? (NULL: all)
	const int32_t *q_start, *q_end;   // query bounds per read as ConClave left them (already counted from the end of the read where
	                             // it filed the reverse complement, conclave.c:131-146), NULL: whole reads
	int M, MM, U, W1, Wl;
	int d[25];
	int minlen, mq;
	double scoreT, mrc;
	// scratch, lane-interleaved unless noted
	int32_t *s32;                // MEM arrays (7 x cap1)
	int32_t *rows;               // 4 DP rows of ncols ints
	uint8_t *E;                  // e_cap bytes per lane, contiguous per lane
	uint32_t *ops_s;             // ops_cap run slots
	int64_t lanes, e_cap;
	int mem_cap, ncols, ops_cap;
	// out
	int32_t *o_stats;            // 10 per read: score, start, end, aln_len, clip_start, clip_end, match, tGaps, qGaps, mapQ
	int64_t *o_off;              // per read: first run in `ops`
	int32_t *o_nops;
	uint32_t *ops;               // runs: (length << 2) | class, class 0 '=' 1 'X' 2 'I' 3 'D'
	int64_t ops_pool_cap;
	unsigned long long *counters;   // [0] run pool top, [1] status, [3] reads put off
	// Two passes. fast != 0: a read whose DP problems all have a provably ungapped answer (see diag_proof) is finished without
	// a move matrix; any other read is put off -- its index goes to q_out -- so that no lane of a wave sits through another
	// lane's matrices. The second pass (q_in = that list, fast = 0) gives the reads put off the full treatment.
	// Three passes in all: between the two, reads whose matrices are small get them in LDS (lds_bytes per lane: four rows of
	// lds_ncols ints, the rest move bytes; 64 lanes per workgroup) -- a lane alone with its matrix in HBM waits a microsecond per
	// cell, and the kernel lasts as long as its slowest lane. What does not fit LDS is put off once more.
	const int32_t *q_in; int32_t *q_out;
	int q_in_cnt, q_out_cnt;        // counter words holding the length of q_in / q_out
	int fast, gap_m_max, ts;
	int lds_bytes, lds_ncols;
};

struct Emit {
	uint32_t *ops; int64_t stride; int cap, n;
	bool over;
	__device__ __forceinline__ uint32_t &at(int i) const { return ops[(int64_t) i * stride]; }
	__device__ void push(int cls, int len) {
		if(len <= 0) return;
		if(n && (int) (at(n - 1) & 3u) == cls) { at(n - 1) += (uint32_t) len << 2; return; }
		if(n < cap) at(n++) = ((uint32_t) len << 2) | (uint32_t) cls; else over = true;
	}
	// drop the last `cols` columns
	__device__ void pop(int cols) {
		while(cols > 0 && n) {
			const int have = (int) (at(n - 1) >> 2);
			if(have <= cols) { cols -= have; --n; }
			else { at(n - 1) -= (uint32_t) cols << 2; cols = 0; }
		}
	}
	__device__ int last_cls() const { return n ? (int) (at(n - 1) & 3u) : -1; }
	__device__ int last_len() const { return n ? (int) (at(n - 1) >> 2) : 0; }
};

struct TLane {
	Lane L;
	int32_t *rows; uint8_t *E; int64_t e_cap;
	Emit em;
	int status;     // 1: a DP problem did not fit the per-lane move matrix; 32: put off to the second pass
	int fast, gap_m_max;
	int ts;                    // -ts: bases trimmed off the front of the chain's seeds (trimSeeds, chain.c:493-528)
	int ncols;                 // DP columns the rows hold
	int64_t row_stride;        // distance between neighbouring row elements (the lane count in HBM, 1 in a lane's own LDS)
};

// DP problems whose answer is known without the matrix. With match M > 0 > mismatch MM > gap open W1, extension U < 0 (checked
// on the host before `fast` is set), an ungapped alignment of g read bases with m mismatches scores (g - m) M + m MM.
//  * Between two seeds (NW with both ends fixed, t_l == q_l == g): any other alignment holds a gap in the read and a gap in the
//    template and at most g - 1 pairs: <= (g - 1) M + 2 W1. The diagonal is the only optimum when m (M - MM) < M - 2 W1
//    (m <= gap_m_max; 2 with the defaults).
//  * A tail next to a seed (read end fixed, template end free, k = -1 / 1), m <= 1: the read base next to the seed is paired
//    with its own template base (then a gap elsewhere: <= MM + (g - 1) M + W1), with a gap (<= W1 + (g - 1) M), or with an
//    earlier template base after a gap of 1 (exactly the diagonal shifted by one: W1 + (g - x1) M + x1 MM for x1 mismatches on
//    it, or <= 2 W1 + g M with a further gap) or of 2 and more (<= W1 + U + g M). With MM > W1, MM - M > 2 W1 and
//    MM - M > W1 + U all of these are below (g - 1) M + MM unless x1 == 0.
// Being the only optimum, the diagonal is what the reference's traceback yields whatever its tie rules.
__device__ __forceinline__ void diag_emit(TLane &T, const uint64_t *ts, const QView &q, int tp, int qp, int g, Aln &r) {
	int score = 0, cls = -1, run = 0;
	for(int i = 0; i < g; ++i) {
		const int tb = tn(ts, tp + i), qb = qn(q, qp + i);
		score += T.L.d[5 * tb + qb];
		const int c = tb == qb ? 0 : 1;
		if(c != cls) { T.em.push(cls, run); cls = c; run = 0; }
		++run;
	}
	T.em.push(cls, run);
	r.score = score; r.len = g; r.match = g; r.tGaps = 0; r.qGaps = 0; r.pos = 0;
}

#define TROW(T, r, n) (T).rows[((int64_t) ((r) * (T).ncols + (n))) * (T).row_stride]

// walk of the move matrix (nw.c:256-305 / :586-635), emitting columns. stride = bytes per template row, dn = column
// change per template step (0 full, -1 band), q_pos = query index of the start column. lead_trim: gaps in front are
// dropped (leadTailAln with t_s == 0, align.c:97-112): a dropped gap-in-template column leaves a read base unaligned.
__device__ void trace_walk(TLane &T, const uint8_t *E, int stride, int m, int n, int dn, int q_pos, const uint64_t *ts,
                           int nuc_pos, int tlen_total, const QView &q, int q_s, int q_len, bool lead_trim, Aln &s, int &clip_start, int &clip_end) {
	const uint8_t *row = E + (int64_t) m * stride;
	s.len = s.match = s.tGaps = s.qGaps = 0;
	bool lead = lead_trim;
	while(row[n] != 0) {
		if(nuc_pos == tlen_total) nuc_pos = 0;
		const int mv = row[n] & 7;
		if(mv == 1) {
			const int tb = tn(ts, nuc_pos), qb = qn(q, q_s + q_pos);
			T.em.push(tb == qb ? 0 : 1, 1);
			lead = false;
			++s.match; ++s.len; ++nuc_pos; row += stride; n += 1 + dn; ++q_pos;
		} else if(mv >= 4) {
			// gap in the read: template bases consumed until a cell that may open the gap
			int g = 1;
			while(!(row[n] >> 4)) { row += stride; n += dn; ++g; }
			row += stride; n += dn;
			nuc_pos += g;        // like the reference, the wrap of a circular template is only taken on a column boundary
			if(!lead) { T.em.push(3, g); s.qGaps += g; s.len += g; }
		} else {
			int g = 1;
			while(!(row[n] >> 3)) { ++n; ++g; }
			++n;
			q_pos += g;
			if(!lead) { T.em.push(2, g); s.tGaps += g; s.len += g; } else clip_start += g;
		}
	}
	clip_end = q_len - q_pos;
}

// NW / NW_band with the move matrix (nw.c:26-309, 310-640; fill identical to NW_score / NW_band_score).
// band < 0: full matrix. Returns false if E does not fit.
__device__ bool nw_trace(TLane &T, const uint64_t *ts, int tlen_total, const QView &q, int k, int t_s, int t_e, int q_s, int q_e,
                         int band, bool lead_trim, int circ_pos, Aln &s, int &clip_start, int &clip_end) {
	const Lane &L = T.L;
	const int W1 = L.W1, U = L.U;
	int t_len = t_e - t_s;
	const int q_len = q_e - q_s;
	if(t_len < 0) t_len += tlen_total;
	s.pos = 0; clip_start = 0; clip_end = 0;
	if(t_len == 0 || q_len == 0) {
		s = nw_degenerate(t_len, q_len, U, W1);
		if(t_len == 0) T.em.push(2, q_len); else T.em.push(3, t_len);
		(void) circ_pos;
		return true;
	}
	const int low = (t_len + q_len) * (L.MM + U + W1);
	if(band < 0) {
		const int pitch = q_len + 1;
		if((int64_t) pitch * (t_len + 1) > T.e_cap || q_len + 2 > T.ncols) return false;
		uint8_t *E = T.E, *Er = E + (int64_t) pitch * t_len;
		int dc = 0, dp = 1, pc = 2, pp = 3;
		s.score = low;
		for(int m = 0; m < t_len; ++m) E[(int64_t) pitch * m + q_len] = (0 < k) ? 0 : 5;
		if(!(0 < k)) E[(int64_t) pitch * (t_len - 1) + q_len] = 36;
		if(k == 2) {
			for(int n = q_len; n >= 0; --n) { TROW(T, dp, n) = 0; TROW(T, pp, n) = low; Er[n] = 0; }
		} else {
			for(int n = q_len - 1; n >= 0; --n) { TROW(T, dp, n) = W1 + (q_len - 1 - n) * U; TROW(T, pp, n) = low; Er[n] = 3; }
			Er[q_len - 1] = 18; Er[q_len] = 0; TROW(T, dp, q_len) = 0; TROW(T, pp, q_len) = 0;
		}
		int best_m = 0, npos = t_e - 1;
		for(int m = t_len - 1; m >= 0; --m, --npos) {
			if(npos < 0) npos = tlen_total - 1;
			uint8_t *e = E + (int64_t) pitch * m;
			// (the cell to the right in this row and the cell below it stay in registers: a lane alone with its rows in HBM waited
			// a memory round trip per cell for the value it had just stored)
			int Dright = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
			TROW(T, dc, q_len) = Dright;
			int Qprev = low;
			int dp_right = TROW(T, dp, q_len);
			const int tb = tn(ts, npos);
			auto one = [&](int n, int dp_n, int pp_n) {
				uint8_t cell = 0, mv;
				int Q = Dright + W1;
				int P = dp_n + W1;
				int D;
				if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
				int x = Qprev + U;
				if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
				x = pp_n + U;
				if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
				x = dp_right + L.d[5 * tb + qn(q, q_s + n)];
				if(D <= x) { D = x; cell |= 1; } else cell |= mv;
				TROW(T, dc, n) = D; TROW(T, pc, n) = P; e[n] = cell; Qprev = Q;
				Dright = D; dp_right = dp_n;
			};
			int n = q_len - 1;
			for(; n >= 3; n -= 4) {
				// the row below, four cells at a time: eight loads in flight instead of two
				const int a0 = TROW(T, dp, n), a1 = TROW(T, dp, n - 1), a2 = TROW(T, dp, n - 2), a3 = TROW(T, dp, n - 3);
				const int b0 = TROW(T, pp, n), b1 = TROW(T, pp, n - 1), b2 = TROW(T, pp, n - 2), b3 = TROW(T, pp, n - 3);
				one(n, a0, b0); one(n - 1, a1, b1); one(n - 2, a2, b2); one(n - 3, a3, b3);
			}
			for(; n >= 0; --n) one(n, TROW(T, dp, n), TROW(T, pp, n));
			if(k < 0 && s.score < TROW(T, dc, 0)) { s.score = TROW(T, dc, 0); best_m = m; }
			int x = dc; dc = dp; dp = x; x = pc; pc = pp; pp = x;
		}
		int sm = 0, sn = 0;
		if(k < 0) {
			sm = best_m;
			if(k == -2) for(int n = 0; n < q_len; ++n) if(s.score <= TROW(T, dp, n)) { s.score = TROW(T, dp, n); sm = 0; sn = n; }
		} else s.score = TROW(T, dp, 0);
		const int score = s.score;
		clip_start = sn;
		trace_walk(T, E, pitch, sm, sn, 0, sn, ts, sm + t_s, tlen_total, q, q_s, q_len, lead_trim, s, clip_start, clip_end);
		s.score = score; s.pos = 0;
		return true;
	}
	// banded (nw.c:310-640)
	if(band & 1) ++band;
	const int half = band >> 1, bq = band + 1, pitch = bq + 1;
	if((int64_t) pitch * (t_len + 1) > T.e_cap || band + 4 > T.ncols) return false;
	uint8_t *E = T.E, *Er = E + (int64_t) pitch * t_len;
	int dc = 0, dp = 1, pc = 2, pp = 3;
	s.score = low;
	int c = (t_len + q_len) >> 1;
	int sn = q_len - 1 - (c - half);
	if(k != 2) {
		for(int n = sn - 1; n >= 0; --n) { TROW(T, dp, n) = W1 + (sn - n - 1) * U; TROW(T, pp, n) = low; Er[n] = 3; }
		Er[sn - 1] = 18; Er[sn] = 0; TROW(T, dp, sn) = 0; TROW(T, pp, sn) = 0;
	} else {
		for(int n = sn; n >= 0; --n) { TROW(T, dp, n) = 0; TROW(T, pp, n) = low; Er[n] = 0; }
	}
	int bm = 0, bn = 0, en = 0, n = 0, npos = t_e - 1;
	for(int m = t_len - 1; m >= 0; --m, --npos, --c) {
		if(npos < 0) npos = tlen_total - 1;
		uint8_t *e = E + (int64_t) pitch * m;
		int sq = c + half, eq = c - half;
		if(eq < 0) { eq = 0; ++en; } else en = 0;
		int Qprev = low;
		if(sq < q_len - 1) {
			sn = bq - 1; TROW(T, dc, bq) = low; e[bq] = 37;
		} else {
			sq = q_len - 1; sn = en + (q_len - eq);
			TROW(T, dc, sn) = (0 < k) ? 0 : (W1 + (t_len - 1 - m) * U);
			e[sn] = (0 < k) ? 0 : 37;
			--sn;
		}
		const int tb = tn(ts, npos);
		int qp = sq;
		int Dright = TROW(T, dc, sn + 1), dp_here = TROW(T, dp, sn);      // (kept in registers from one cell to the next, as above)
		for(n = sn; n > en; --qp, --n) {
			uint8_t cell = 0, mv;
			const int dp_left = TROW(T, dp, n - 1), pp_left = TROW(T, pp, n - 1);
			int Q = Dright + W1;
			int P = dp_left + W1;
			int D;
			if(Q < P) { D = P; mv = 4; } else { D = Q; mv = 2; }
			int x = Qprev + U;
			if(Q < x) { Q = x; if(D <= x) { D = x; mv = 3; } } else cell |= 16;
			x = pp_left + U;
			if(P < x) { P = x; if(D <= x) { D = x; mv = 5; } } else cell |= 32;
			x = dp_here + L.d[5 * tb + qn(q, q_s + qp)];
			if(D <= x) { D = x; cell |= 1; } else cell |= mv;
			TROW(T, dc, n) = D; TROW(T, pc, n) = P; e[n] = cell; Qprev = Q;
			Dright = D; dp_here = dp_left;
		}
		{	// band edge: no gap-in-query state (nw.c:1079-1105)
			uint8_t cell = 0, mv;
			int Q = Dright + W1, x = Qprev + U;
			if(Q < x) { Q = x; mv = 3; } else { mv = 2; cell |= 16; }
			TROW(T, pc, n) = low;
			int D = dp_here + L.d[5 * tb + qn(q, q_s + qp)];
			if(Q <= D) cell |= 1; else { D = Q; cell |= mv; }
			TROW(T, dc, n) = D; e[n] = cell;
		}
		if(eq == 0 && k < 0 && s.score < TROW(T, dc, n)) { s.score = TROW(T, dc, n); bm = m; bn = n; }
		int x = dc; dc = dp; dp = x; x = pc; pc = pp; pp = x;
	}
	int q_pos = 0;
	if(bm == 0) { bn = en; s.score = TROW(T, dp, en); }
	if(k == -2) for(n = en; n < bq; ++n) if(s.score <= TROW(T, dp, n)) { s.score = TROW(T, dp, n); bm = 0; bn = n; q_pos = n - en; }
	const int score = s.score;
	clip_start = q_pos;
	trace_walk(T, E, pitch, bm, bn, -1, q_pos, ts, bm + t_s, tlen_total, q, q_s, q_len, lead_trim, s, clip_start, clip_end);
	s.score = score; s.pos = 0;
	return true;
}

// KMA(), align.c:214-507. Returns the alignment statistics (len == 1, score == 0: no alignment); columns go to T.em.
// FAST (the first pass): problems are either settled on the diagonal or the read is put off -- no move matrix in this instantiation,
// which is what lets it run at more waves per SIMD than the full one
template <bool FAST>
__device__ Aln kma_trace(TLane &T, const DevDB &db, int t, const uint64_t *ts, int t_len, const QView &q, int mq,
                         int &clip_start, int &clip_end, unsigned &mapQ) {
	const Aln FAIL = {0, 1, 0, 0, 0, 0};
	const Lane &L = T.L;
	const int k = (int) db.kmersize, q_len = q.L, bw = 64, cap = L.cap1 - 1;
	clip_start = clip_end = 0; mapQ = 0;
	T.em.n = 0;
	int nm = 0;
	// seeds: the byte-wise loop of KMA() -- a stretch between Ns (or up to the read end) is probed only while MORE than
	// k bases remain in it (align.c:258, 308, 364), unlike KMA_score
	{
		int i = q.b0, ni = 1;
		const int q_stop = qb1(q);           // query bounds: align.c:249-254
		while(i < q_stop) {
			while(ni <= q.nN && qN_at(q, ni) < i) ++ni;
			const int end = (ni <= q.nN) ? qN_at(q, ni) : q_stop;
			const int lowq = (ni > 1) ? qN_at(q, ni - 1) + 1 : 0;
			if(i < end - k) i += k - 1; else { i = end + 1; continue; }
			while(i < end) {
				const int v = tpos_get(db, t, q_kmer(q, i - (k - 1), k));
				if(v == 0) { ++i; continue; }
				i -= k - 1;
				if(v > 0) {
					if(nm >= cap) { T.status = 2; return FAIL; }
					i = add_mem(L, nm, ts, t_len, q, i, v, k, lowq, end);
					++nm;
				} else {
					const int32_t *dl = db.tpos_dups + (-v - 1);
					const int cnt = dl[0];
					int bias = i;
					for(int c = 1; c <= cnt; ++c) {
						if(nm >= cap) { T.status = 2; return FAIL; }
						const int qe = add_mem(L, nm, ts, t_len, q, i, dl[c], k, lowq, end);
						++nm;
						bias = max(bias, qe);
					}
					i = bias + 1;
				}
				if(i < end - k) i += k - 1; else i = end + 1;
			}
			i = end + 1;
		}
	}
	if(!nm) return FAIL;
	int start = chain_seeds(L, nm, q_len, t_len, k, &mapQ);
	if(mapQ < (unsigned) mq || MEMA(L, 5, start) < k) return FAIL;
	if(T.ts) {
		// trimSeeds (chain.c:493-528; align.c:413): the front of every seed of the chain -- but not of a first seed that starts the read --
		// is given back to the DP problem before it, one base of a seed stays at least
		for(int c = MEMA(L, 2, start) ? start : MEMA(L, 6, start); c; c = MEMA(L, 6, c)) {
			const int len = MEMA(L, 3, c) - MEMA(L, 2, c);
			const int cut = len < T.ts ? len - 1 : T.ts;
			MEMA(L, 0, c) += cut; MEMA(L, 2, c) += cut;
		}
	}

	Aln S = {0, 0, 0, 0, 0, 0};
	{	// leading tail (leadTailAln with Frag_align, align.c:53-131)
		const int t_e = MEMA(L, 0, start) - 1, q_e = MEMA(L, 2, start);
		S.pos = t_e;
		if(q_e) {
			int t_s = 0, q_s = 0;
			if((q_e << 1) < t_e || (q_e + bw) < t_e) t_s = t_e - (q_e + (q_e < bw ? q_e : bw));
			else if((t_e << 1) < q_e || (t_e + bw) < q_e) q_s = q_e - (t_e + (t_e < bw ? t_e : bw));
			if(t_e - t_s > 0 && q_e - q_s > 0) {
				const int band = abs(t_e - t_s - q_e + q_s) + bw;
				const bool full = q_e - q_s <= band || t_e - t_s <= band;
				Aln r; int cs = 0, ce = 0;
				if(FAST) {
					const int g = q_e - q_s;
					if(t_s == 0 || !full || q.nN) { T.status = 32; return FAIL; }
					const int m = diag_mism(ts, q, t_e - g, q_s, g);
					if(m > 1 || (m == 1 && diag_mism(ts, q, t_e - 1 - g, q_s, g) == 0)) { T.status = 32; return FAIL; }
					diag_emit(T, ts, q, t_e - g, q_s, g, r);
				} else if(!nw_trace(T, ts, t_len, q, -1 - (t_s == 0), t_s, t_e, q_s, q_e, full ? -1 : band, t_s == 0, 0, r, cs, ce)) { T.status = 1; return FAIL; }
				clip_start = q_s + cs;
				S.pos -= r.len - r.tGaps;
				S.score = r.score; S.len = r.len; S.match = r.match; S.tGaps = r.tGaps; S.qGaps = r.qGaps;
			} else clip_start = q_s;
		}
	}
	for(;;) {
		const int qS = MEMA(L, 2, start), qE = MEMA(L, 3, start);
		T.em.push(0, qE - qS);
		S.len += qE - qS; S.match += qE - qS;
		for(int i = qS; i < qE; ++i) { const int b = qn(q, i); S.score += L.d[6 * b]; }
		const int nxt = MEMA(L, 6, start);
		if(!nxt) break;
		const int q_s = qE, t_s = MEMA(L, 1, start) - 1;
		start = nxt;
		int qSn = MEMA(L, 2, start), tSn = MEMA(L, 0, start);
		if(qSn < q_s) { tSn += q_s - qSn; qSn = q_s; }
		int t_e = tSn - 1, t_l;
		if(t_e < t_s) {
			if(t_s <= MEMA(L, 1, start)) { qSn += t_s - t_e; t_e = t_s; t_l = 0; }
			else t_l = t_len - t_s + t_e;
		} else t_l = t_e - t_s;
		MEMA(L, 2, start) = qSn; MEMA(L, 0, start) = tSn;
		const int q_e = qSn;
		if(abs(t_l - q_e + q_s) * L.U > q_len * L.M || t_l > q_len || q_e - q_s > (q_len >> 1)) return FAIL;
		if(t_l > 0 || q_e - q_s > 0) {
			const int band = abs(t_l - q_e + q_s) + bw;
			const bool full = q_e - q_s <= band || t_l <= band;
			Aln r; int cs, ce;
			if(FAST && t_l > 0 && q_e - q_s > 0) {
				if(t_l != q_e - q_s || t_e < t_s || !full || q.nN || diag_mism(ts, q, t_s, q_s, t_l) > T.gap_m_max) { T.status = 32; return FAIL; }
				diag_emit(T, ts, q, t_s, q_s, t_l, r);
			} else if(FAST) {
				// one side empty: a single gap run, no matrix (the head of NW, nw.c:37-60)
				const int ql = q_e - q_s;
				r = nw_degenerate(t_l, ql, L.U, L.W1);
				if(t_l == 0) T.em.push(2, ql); else T.em.push(3, t_l);
			} else if(!nw_trace(T, ts, t_len, q, 0, t_s, t_e, q_s, q_e, full ? -1 : band, false, t_len, r, cs, ce)) { T.status = 1; return FAIL; }
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
	}
	{	// trailing tail (trailTailAln with Frag_align, align.c:140-212)
		const int t_s = MEMA(L, 1, start) - 1, q_s = MEMA(L, 3, start);
		int q_e = q_len, t_e = t_len, fr_end = 0;
		if(((q_len - q_s) << 1) < (t_len - t_s) || (q_len - q_s + bw) < (t_len - t_s)) {
			t_e = q_len - q_s; t_e = t_s + (t_e + (t_e < bw ? t_e : bw));
		} else if(((t_len - t_s) << 1) < (q_len - q_s) || (t_len - t_s + bw) < (q_len - q_s)) {
			q_e = t_len - t_s; q_e = q_s + (q_e + (q_e < bw ? q_e : bw));
		}
		if(t_e - t_s > 0 && q_e - q_s > 0) {
			const int band = abs(t_e - t_s - q_e + q_s) + bw;
			const bool full = q_e - q_s <= band || t_e - t_s <= band;
			Aln r; int cs, ce = 0;
			if(FAST) {
				const int g = q_e - q_s;
				if(t_e == t_len || !full || q.nN) { T.status = 32; return FAIL; }
				const int m = diag_mism(ts, q, t_s, q_s, g);
				if(m > 1 || (m == 1 && diag_mism(ts, q, t_s + 1, q_s, g) == 0)) { T.status = 32; return FAIL; }
				diag_emit(T, ts, q, t_s, q_s, g, r);
			} else if(!nw_trace(T, ts, t_len, q, 1 + (t_e == t_len), t_s, t_e, q_s, q_e, full ? -1 : band, false, 0, r, cs, ce)) { T.status = 1; return FAIL; }
			fr_end = ce;
			if(t_e == t_len) {
				// gaps at the very end of the template are trimmed; the first column of the tail always stays
				int left = r.len - 1;
				while(left > 0 && T.em.last_cls() >= 2) {
					const int take = min(left, T.em.last_len()), cls = T.em.last_cls();
					T.em.pop(take);
					if(cls == 2) { r.tGaps -= take; fr_end += take; } else r.qGaps -= take;
					r.len -= take; left -= take;
				}
			}
			S.score += r.score; S.len += r.len; S.match += r.match; S.tGaps += r.tGaps; S.qGaps += r.qGaps;
		}
		clip_end = q_len - q_e + fr_end;
	}
	return S;
}

template <bool FAST>
__global__ __launch_bounds__(256) void trace_kernel(const TraceArgs A) {      // (64 threads in the LDS pass)
	__shared__ int s_d[25];
	if(threadIdx.x < 25) s_d[threadIdx.x] = A.d[threadIdx.x];
	__syncthreads();
	const int64_t gtid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(gtid >= A.lanes) return;
	TLane T;
	Lane &L = T.L;
	L.s32 = A.s32 + gtid; L.s64 = nullptr; L.r32 = nullptr; L.r64 = nullptr; L.lanes = A.lanes; L.cap1 = A.mem_cap + 1; L.ncols = A.ncols;
	L.d = s_d; L.M = A.M; L.MM = A.MM; L.U = A.U; L.W1 = A.W1;
	L.cnt = nullptr; L.wide = nullptr; L.queue = nullptr; L.xq = nullptr; L.xq_cnt = nullptr; L.xq_cap = 0; L.q_at = 0; L.q_mate = 0; L.q_rd = 0; L.ablate = 0; L.gap_m_max = -1;
	L.diag_uniform = 0;
	T.rows = A.rows + gtid; T.E = A.E + gtid * A.e_cap; T.e_cap = A.e_cap; T.ncols = A.ncols; T.row_stride = A.lanes;
	if(A.lds_bytes) {
		extern __shared__ __align__(16) uint8_t t_lds[];
		uint8_t *mine = t_lds + (size_t) threadIdx.x * A.lds_bytes;
		T.rows = (int32_t *) mine; T.ncols = A.lds_ncols; T.row_stride = 1;
		T.E = mine + 16 * A.lds_ncols; T.e_cap = A.lds_bytes - 16 * A.lds_ncols;
	}
	T.em.ops = A.ops_s + gtid; T.em.stride = A.lanes; T.em.cap = A.ops_cap;
	T.fast = A.fast; T.gap_m_max = A.gap_m_max; T.ts = A.ts;
	const int lane = threadIdx.x & 63;
	const int64_t n_items = A.q_in ? (int64_t) A.counters[A.q_in_cnt] : A.n_reads;       // (the list was filled by the launch before this one)
	const int64_t n_threads = (int64_t) gridDim.x * blockDim.x;
	for(int64_t it = gtid; __any(it < n_items); it += n_threads) {
		const int64_t r = it < n_items ? (A.q_in ? (int64_t) A.q_in[it] : it) : A.n_reads;
		// what the read contributes: nothing (keep = false) or its figures + T.em.n alignment runs
		bool keep = false;
		int32_t *st = nullptr;
		int read_score = 0, start = 0, end = 0, aln_len = 0, cs = 0, ce = 0, t_len = 0;
		unsigned mapQ = 0;
		Aln S = {0, 0, 0, 0, 0, 0};
		if(r < A.n_reads) {
			st = A.o_stats + 10 * r;
			for(int x = 0; x < 10; ++x) st[x] = 0;
			A.o_off[r] = 0; A.o_nops[r] = 0;
			const int tt = A.tmpl[r], t = abs(tt);
			if(!(t == 0 || (A.tmpl_ok && !A.tmpl_ok[t]))) {
				QView q;
				q.w = A.seq + A.seq_off[r]; q.L = A.len[r]; q.N = A.N + A.N_off[r]; q.nN = (int) (A.N_off[r + 1] - A.N_off[r]);
				q.rc = (((A.flag[r] & 1) != 0) != (tt < 0)) ? 1 : 0;
				q_set_bounds(q, A.q_start, A.q_end, r);
				t_len = A.db.tlen[t];
				const uint64_t *ts = A.db.tseq + A.db.tseq_off[t];
				T.em.n = 0; T.em.over = false; T.status = 0;
				S = kma_trace<FAST>(T, A.db, t, ts, t_len, q, A.mq, cs, ce, mapQ);
				if(T.status == 32 || (T.status == 1 && A.lds_bytes)) A.q_out[atomicAdd(&A.counters[A.q_out_cnt], 1ull)] = (int32_t) r;
				else if(T.status || T.em.over) atomicMax(&A.counters[1], (unsigned long long) (T.em.over ? 4 : (T.status == 1 ? 8 : 16)));
				else {
					// assemble_KMA, assembly.c:1931-1961
					aln_len = S.len; start = S.pos;
					end = start + aln_len - S.tGaps;
					if(t_len < end) end -= t_len;
					read_score = S.score;
					if(start == 0) read_score += A.Wl;
					if(end == t_len) read_score += A.Wl;
					double score = 0;
					if(A.minlen <= aln_len && ((A.mrc * q.L <= S.len - S.qGaps) || (A.mrc * t_len <= S.len - S.tGaps))) score = 1.0 * read_score / aln_len;
					else read_score = 0;
					keep = 0 < read_score && A.scoreT <= score;
				}
			}
		}
		// room in the run pool: one atomic per wavefront (a wave prefix sum of the run counts) instead of one per read on a
		// single counter
		const int need = keep ? T.em.n : 0;
		int incl = need;
		for(int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if(lane >= d) incl += v; }
		const int total = __shfl(incl, 63);
		unsigned long long base = 0;
		if(lane == 63 && total) base = atomicAdd(&A.counters[0], (unsigned long long) total);
		base = __shfl(base, 63);
		if(!keep) continue;
		const int64_t o = (int64_t) base + incl - need;
		if(o + T.em.n > A.ops_pool_cap) { atomicMax(&A.counters[1], 2ull); continue; }
		for(int x = 0; x < T.em.n; ++x) A.ops[o + x] = T.em.at(x);
		st[0] = read_score; st[1] = start; st[2] = (t_len < end) ? end - t_len : end; st[3] = aln_len; st[4] = cs; st[5] = ce;
		st[6] = S.match; st[7] = S.tGaps; st[8] = S.qGaps; st[9] = (int) mapQ;
		A.o_off[r] = o; A.o_nops[r] = T.em.n;
	}
}

} // namespace

// common launcher. SE: one record per read. PE (rec_mate != null): two records per pair over interleaved mates.
// ---- the tasks of long reads (long_routed) through the pipeline of longtrace.hip in KMA_score mode: a task = a read of its own there,
// with its template and strand; what comes back are KMA_score's figures, filtered like alnFragsSE does (alnfrags.c:1127-1168) ----
__global__ __launch_bounds__(256) void long_pick_kernel(const AlignArgs A, int64_t *list, unsigned long long *cnt) {
	const int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(r >= A.n_reads || A.T_off[A.n_reads] > A.tasks_cap || !long_routed(A, r)) return;
	const int64_t a = A.T_off[r], m = A.T_off[r + 1] - a;
	if(m <= 0) return;
	const unsigned long long b = atomicAdd(cnt, (unsigned long long) m);
	for(int64_t j = 0; j < m; ++j) list[b + j] = a + j;
}
__global__ __launch_bounds__(256) void long_gather_kernel(const AlignArgs A, const int64_t *list, int64_t n_l, int64_t *so, int32_t *len, int32_t *tmpl, int32_t *rc, int32_t *qs, int32_t *qe) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n_l) return;
	const int64_t task = list[i], r = A.t_rec[task];
	so[i] = A.seq_off[r]; len[i] = A.len[r]; tmpl[i] = abs(A.T[task]); rc[i] = (A.flag[r] & 16) ? 1 : 0;
	if(qs) { qs[i] = A.q_start[r]; qe[i] = A.q_end[r]; }
}
__global__ __launch_bounds__(256) void long_result_kernel(const AlignArgs A, const int64_t *list, int64_t n_l, const int32_t *stats) {
	const int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(i >= n_l) return;
	const int64_t task = list[i], r = A.t_rec[task];
	const int32_t *st = stats + 10 * i;
	const int t_len = A.db.tlen[abs(A.T[task])], q_len = A.len[r];
	const int alen = st[3], start = st[1], tG = st[7], qG = st[8];
	int end = start + alen - tG;
	if(t_len < end) end -= t_len;
	const double denom = (q_len <= alen || t_len <= alen) ? (double) alen : (double) min(q_len, t_len);
	int rs = st[0];
	double norm = 0.0;
	if(A.minlen <= alen && ((A.mrc * q_len <= alen - qG) || (A.mrc * t_len <= alen - tG))) norm = rs / denom;
	else rs = 0;
	A.t_tmpl[task] = A.T[task];
	A.t_score[task] = rs; A.t_alen[task] = alen; A.t_start[task] = start; A.t_end[task] = end; A.t_norm[task] = norm;
}

static int long_tasks(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const AlignArgs &A, const kmahip_params *p, hipStream_t stream) {
	const int64_t n = reads->n_reads, cap = A.tasks_cap;
	auto t_in = std::chrono::steady_clock::now();
	const int tim_mode = getenv("KMAHIP_ALIGN_LONG_TIMING") ? atoi(getenv("KMAHIP_ALIGN_LONG_TIMING")) : 0;
	if(tim_mode) { if(tim_mode == 1) HIP_TRY(hipStreamSynchronize(stream)); fprintf(stderr, "[kmahip] stage 3a: the kernels before the long reads' tasks %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count()); }
	struct Dev { void *p = nullptr; ~Dev() { if(p) (void) hipFree(p); } } d_list, d_arr;
	HIP_TRY(hipMalloc(&d_list.p, (size_t) cap * 8 + 16));
	int64_t *list = (int64_t *) d_list.p;
	unsigned long long *cnt = (unsigned long long *) (list + cap);
	HIP_TRY(hipMemsetAsync(cnt, 0, 8, stream));
	hipLaunchKernelGGL(long_pick_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, A, list, cnt);
	HIP_TRY(hipGetLastError());
	unsigned long long n_l = 0;
	HIP_TRY(hipMemcpyAsync(&n_l, cnt, 8, hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipStreamSynchronize(stream));
	if(n_l == 0) return KMAHIP_OK;
	if((int64_t) n_l > cap) { kmahip_set_error("long-read tasks: %llu for %lld slots", n_l, (long long) cap); return KMAHIP_EDEVICE; }
	// the batch of tasks: offsets into the same packed reads, no N's (long_routed)
	const size_t m = (size_t) n_l + 1;
	const size_t words = 2 * m /*seq_off*/ + 2 * m /*N_off*/ + 2 * m /*ops_off*/ + 6 * m /*len tmpl rc qs qe n_ops*/ + 10 * m /*stats*/;
	HIP_TRY(hipMalloc(&d_arr.p, words * 4));
	HIP_TRY(hipMemsetAsync(d_arr.p, 0, words * 4, stream));
	int64_t *so = (int64_t *) d_arr.p, *no = so + m, *oo = no + m;
	int32_t *len = (int32_t *) (oo + m), *tmpl = len + m, *rc = tmpl + m, *qs = rc + m, *qe = qs + m, *nops = qe + m, *stats = nops + m;
	const bool bounds = A.q_start != nullptr && A.q_end != nullptr;
	hipLaunchKernelGGL(long_gather_kernel, dim3((unsigned) ((n_l + 255) / 256)), dim3(256), 0, stream, A, list, (int64_t) n_l, so, len, tmpl, rc, bounds ? qs : nullptr, bounds ? qe : nullptr);
	HIP_TRY(hipGetLastError());
	kmahip_reads L{};
	L.n_reads = (int64_t) n_l; L.seq = reads->seq; L.seq_off = so; L.len = len; L.N = reads->N; L.N_off = no; L.seq_words = reads->seq_words; L.N_total = 0; L.max_len = reads->max_len;
	if(bounds) { L.q_start = qs; L.q_end = qe; }
	kmahip_traces tr{};
	tr.stats = stats; tr.ops_off = oo; tr.n_ops = nops; tr.ops = nullptr; tr.ops_cap = 0;
	const bool tim = getenv("KMAHIP_ALIGN_LONG_TIMING") != nullptr;
	auto t0 = t_in;
	if(tim_mode == 1) HIP_TRY(hipStreamSynchronize(stream));
	auto t1 = std::chrono::steady_clock::now();
	const int rcl = kmahip_launch_longtrace(db, ws, &L, tmpl, 0, rc, nullptr, 0, p, &tr, nullptr, stream, 1);
	if(rcl) return rcl;
	if(tim) {
		fprintf(stderr, "[kmahip] stage 3a: %llu tasks of long reads: lists %.1f ms, pipeline %.1f ms\n", n_l, std::chrono::duration<double, std::milli>(t1 - t0).count(),
		        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
	}
	hipLaunchKernelGGL(long_result_kernel, dim3((unsigned) ((n_l + 255) / 256)), dim3(256), 0, stream, A, list, (int64_t) n_l, stats);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(stream));          // (the scratch goes with this call)
	return KMAHIP_OK;
}

static int launch_align(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                        const int32_t *rec_mate, const int32_t *rec_rc, int32_t *pe_kind,
                        const kmahip_params *p, kmahip_hits *out, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	if(n < 0 || !cands || !out || !p) { kmahip_set_error("bad arguments"); return KMAHIP_EINVAL; }
	if(!db->dev.tpos_slots) { kmahip_set_error("index has no .length.b/.seq.b: stage 3a unavailable"); return KMAHIP_EINVAL; }
	if(p->minFrac != 1.0) { kmahip_set_error("minFrac != 1.0 not supported"); return KMAHIP_EINVAL; }
	if(n == 0) return KMAHIP_OK;
	int max_len = reads->max_len;
	if(max_len <= 0) { kmahip_set_error("kmahip_reads.max_len must be set for the align stage"); return KMAHIP_EINVAL; }
	if(max_len > (1 << 20)) { kmahip_set_error("reads longer than 2^20 bases not supported"); return KMAHIP_EINVAL; }
	// scratch geometry
	const int mem_cap = (max_len <= 1024 ? 64 : max_len / 8) * std::max(1, ws->mem_scale);
	const int ncols = max_len + 72;
	const int64_t tasks_cap = cands->T_cap > 0 ? cands->T_cap : 1;
	int64_t lanes = 256ll * 1024;
	const size_t per_lane32 = (size_t) (7 * (mem_cap + 1) + 2 * ncols) * 4, per_lane64 = (size_t) 2 * ncols * 8;
	while(lanes > 4096 && (size_t) lanes * (per_lane32 + per_lane64) > (6ull << 30)) lanes >>= 1;
	if(ws->a_lanes != lanes || ws->a_mem_cap != mem_cap || ws->a_ncols != ncols) {
		(void) hipFree(ws->a_s32); (void) hipFree(ws->a_s64);
		ws->a_s32 = nullptr; ws->a_s64 = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->a_s32, (size_t) lanes * per_lane32));
		HIP_TRY(hipMalloc((void **) &ws->a_s64, (size_t) lanes * per_lane64));
		ws->a_lanes = lanes; ws->a_mem_cap = mem_cap; ws->a_ncols = ncols;
	}
	if(ws->a_task_cap < tasks_cap) {
		(void) hipFree(ws->a_task);
		ws->a_task = nullptr;
		// (per task: seeds 64 B, norm 8, hand-on list 8, eight int columns 32, pending list 4, row of sums 64, class queues 4 x 48 / 4)
		HIP_TRY(hipMalloc((void **) &ws->a_task, (size_t) tasks_cap * (2 * SEEDS * 8 + 8 + 8 + 8 * 4 + 4 + PART_INTS * 4) + DQ_CLASSES * (size_t) (tasks_cap / 4 + 4096) * QENT * 4 + 64));
		ws->a_task_cap = tasks_cap;
	}
	if(!ws->counters) { HIP_TRY(hipMalloc((void **) &ws->counters, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); HIP_TRY(hipMemset(ws->counters, 0, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); }
	HIP_TRY(hipMemsetAsync(ws->counters + 2, 0, 7 * sizeof(unsigned long long), stream));      // [2] .. [8]
	HIP_TRY(hipMemsetAsync(ws->counters + AC_DQ, 0, (DQ_CLASSES + 1) * sizeof(unsigned long long), stream));  // the class queues and the pending list

	AlignArgs A;
	A.db = db->dev;
	A.n_reads = n; A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.rc_flag = cands->rc_flag; A.flag = cands->flag; A.T_off = cands->T_off; A.T = cands->T;
	A.q_start = reads->q_start; A.q_end = reads->q_end;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1;
	for(int i = 0; i < 5; ++i) for(int j = 0; j < 5; ++j) A.d[i * 5 + j] = p->rw.d[i][j];
	A.minlen = p->minlen; A.mq = p->mq; A.scoreT = p->scoreT; A.mrc = p->mrc;
	// per task: the MEMs of two seed slots first (32 bytes each: read as two 16-byte loads), the norm, the hand-on list, eight int columns
	uint2 *seed_mem = (uint2 *) ws->a_task;
	double *norm = (double *) (seed_mem + (size_t) 2 * SEEDS * tasks_cap);
	A.slow_list = (int64_t *) (norm + tasks_cap); A.use_list = 0; A.per_round = 64;
	A.part = (int *) (A.slow_list + tasks_cap);                         // (rows of 64 bytes, 16-byte aligned: everything before is)
	A.dq_cap = std::min<int64_t>(tasks_cap / 4 + 4096, 1ll << 26);
	A.dq = A.part + (size_t) PART_INTS * tasks_cap;
	int32_t *ti = (int32_t *) (A.dq + DQ_CLASSES * (size_t) (tasks_cap / 4 + 4096) * QENT);
	A.pend = ti + 8 * tasks_cap;
	A.t_norm = norm; A.t_score = ti; A.t_alen = ti + tasks_cap; A.t_start = ti + 2 * tasks_cap; A.t_end = ti + 3 * tasks_cap; A.t_tmpl = ti + 4 * tasks_cap;
	int32_t *t_rec = ti + 5 * tasks_cap;
	A.seed_slots = rec_mate ? 2 : 1;
	A.seed_n = ti + 6 * tasks_cap; A.seed_mem = seed_mem;
	A.t_rec = t_rec;
	if(n >= 0x7FFFFFFF) { kmahip_set_error("too many records in one batch"); return KMAHIP_EINVAL; }
	A.s32 = ws->a_s32; A.s64 = ws->a_s64; A.lanes = lanes; A.mem_cap = mem_cap; A.ncols = ncols;
	A.rec_mate = rec_mate; A.rec_rc = rec_rc; A.pe_mode = rec_mate != nullptr; A.Wl = -p->rw.Wl; A.PE = p->rw.PE;
	// long reads: a 10 kb read has hundreds of link problems per task; what the LDS queues cannot take is spilled per wave
	A.xq = nullptr; A.xq_cap = 0;
	if(max_len > 1024) {
		const int64_t waves = lanes / 64;
		int64_t cap = 16ll * (mem_cap + 2);
		while(cap > 256 && waves * 4 * cap * QENT * 4 > (2ll << 30)) cap >>= 1;
		const size_t bytes = (size_t) waves * 4 * cap * QENT * 4;
		if(ws->a_xq_bytes < bytes) {
			(void) hipFree(ws->a_xq); ws->a_xq = nullptr; ws->a_xq_bytes = 0;
			HIP_TRY(hipMalloc((void **) &ws->a_xq, bytes));
			ws->a_xq_bytes = bytes;
		}
		A.xq = ws->a_xq; A.xq_cap = (int) cap;
	}
	A.counters = ws->counters;
	A.stats = ws->stats_on;
	A.ablate = 0;
	{	// nw_diagonal: only with a plain match / mismatch matrix and penalties ordered as its proof needs (KMAHIP_ALIGN_DIAG=0: off)
		const int M = p->rw.M, MM = p->rw.MM, W1 = p->rw.W1, U = p->rw.U;
		bool plain = M > 0 && MM < 0 && W1 < 0 && U < 0 && MM > W1 && MM - M > 2 * W1 && MM - M > W1 + U;
		for(int i = 0; i < 4 && plain; ++i) for(int j = 0; j < 4; ++j) plain = plain && p->rw.d[i][j] == (i == j ? M : MM);
		const char *e = getenv("KMAHIP_ALIGN_DIAG");
		if(e && !atoi(e)) plain = false;
		A.gap_m_max = plain ? (M - 2 * W1 - 1) / (M - MM) : -1;
	}
#ifdef KMAHIP_DIAG
	if(const char *e = getenv("KMAHIP_ABLATE_ALIGN")) A.ablate = atoi(e);
#endif
	A.tasks_cap = tasks_cap;
	// reads over 1 kb (single end, not the work-counting launch) go through the long-read pipeline (KMAHIP_ALIGN_LONG: another length, 0: never)
	A.long_min = 0;
	if(!rec_mate && !A.stats) {
		const int lm = getenv("KMAHIP_ALIGN_LONG") ? atoi(getenv("KMAHIP_ALIGN_LONG")) : 1025;
		if(lm > 0 && max_len >= lm) A.long_min = std::max(lm, (int) db->dev.kmersize + 1);
	}
	hipLaunchKernelGGL(task_map_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, cands->T_off, n, t_rec, tasks_cap, ws->counters);
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if(A.seed_n && !A.stats) {
		hipEvent_t es0 = nullptr, es1 = nullptr;
		if(ws->timing_on) { HIP_TRY(hipEventCreate(&es0)); HIP_TRY(hipEventCreate(&es1)); HIP_TRY(hipEventRecord(es0, stream)); }
		hipLaunchKernelGGL(seed_tasks_kernel, dim3(256 * 8), dim3(256), 0, stream, A);     // 8 waves / SIMD on 256 CUs
		if(ws->timing_on) {
			HIP_TRY(hipEventRecord(es1, stream));
			if(!ws->events4) ws->events4 = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
			ws->events4->push_back({es0, es1});
		}
	}
	if(ws->timing_on) {
		HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
		HIP_TRY(hipEventRecord(ev0, stream));
	}
	const dim3 agrid((unsigned) (lanes / ATHREADS));
	// the register-only kernel for the tasks seed_tasks_kernel has seeded, the general one over what it hands on (KMAHIP_ALIGN_FAST=0: the
	// general kernel for every task); not for the work-counting launch, nor with a matrix whose diagonal is not uniform
	bool fast = A.seed_n && !A.stats && A.d[0] == A.d[6] && A.d[0] == A.d[12] && A.d[0] == A.d[18] && tasks_cap < (1ll << 31);
	if(const char *e = getenv("KMAHIP_ALIGN_FAST")) if(!atoi(e)) fast = false;
	if(A.stats) {
		if(A.pe_mode) hipLaunchKernelGGL((align_tasks_kernel<true, true>), agrid, dim3(ATHREADS), 0, stream, A);
		else hipLaunchKernelGGL((align_tasks_kernel<true, false>), agrid, dim3(ATHREADS), 0, stream, A);
	} else {
		hipStream_t gstream = stream;
		if(fast) {
			const dim3 fgrid(256 * 8);
			if(A.pe_mode) hipLaunchKernelGGL((align_fast_kernel<true>), fgrid, dim3(FTHREADS), 0, stream, A);
			else hipLaunchKernelGGL((align_fast_kernel<false>), fgrid, dim3(FTHREADS), 0, stream, A);
			// the general kernel over the handed-on tasks (a few wavefronts, each a chain of dependent round trips) beside the class
			// queues: neither reads what the other writes (a task is pending or handed on, never both: see fast_flush)
			if(!ws->a_side && !getenv("KMAHIP_ALIGN_SERIAL")) {
				HIP_TRY(hipStreamCreateWithFlags(&ws->a_side, hipStreamNonBlocking));
				HIP_TRY(hipEventCreateWithFlags(&ws->a_ev[0], hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&ws->a_ev[1], hipEventDisableTiming));
			}
			if(ws->a_side && !getenv("KMAHIP_ALIGN_SERIAL")) {
				HIP_TRY(hipEventRecord(ws->a_ev[0], stream));
				HIP_TRY(hipStreamWaitEvent(ws->a_side, ws->a_ev[0], 0));
				gstream = ws->a_side;
			}
			// the tasks handed on all bring work for the wave's queues; 64 of them in a round would overrun them (8 wide, 20 narrow, 32 tiny
			// problems) and what does not fit is walked by single lanes: fewer tasks per round, all 64 lanes on their problems
			A.use_list = 1; A.per_round = getenv("KMAHIP_ALIGN_ROUND") ? std::min(64, std::max(1, atoi(getenv("KMAHIP_ALIGN_ROUND")))) : 16;
		}
		if(A.pe_mode) hipLaunchKernelGGL((align_tasks_kernel<false, true>), agrid, dim3(ATHREADS), 0, gstream, A);
		else hipLaunchKernelGGL((align_tasks_kernel<false, false>), agrid, dim3(ATHREADS), 0, gstream, A);
		if(fast) {
			const dim3 qgrid(1024);
			hipLaunchKernelGGL((dp_queue_kernel<8>), qgrid, dim3(256), 0, stream, A, 2);
			hipLaunchKernelGGL((dp_queue_kernel<16>), qgrid, dim3(256), 0, stream, A, 1);
			hipLaunchKernelGGL((dp_queue_kernel<32>), qgrid, dim3(256), 0, stream, A, 3);
			hipLaunchKernelGGL((dp_queue_kernel<64>), qgrid, dim3(256), 0, stream, A, 0);
			if(A.pe_mode) hipLaunchKernelGGL((pend_finish_kernel<true>), dim3(512), dim3(256), 0, stream, A);
			else hipLaunchKernelGGL((pend_finish_kernel<false>), dim3(512), dim3(256), 0, stream, A);
			if(gstream != stream) {
				HIP_TRY(hipEventRecord(ws->a_ev[1], gstream));
				HIP_TRY(hipStreamWaitEvent(stream, ws->a_ev[1], 0));
			}
		}
		if(fast && getenv("KMAHIP_DEBUG_TIMING")) {
			unsigned long long c[7];
			HIP_TRY(hipStreamSynchronize(stream));
			HIP_TRY(hipMemcpy(c, ws->counters + 2, sizeof c, hipMemcpyDeviceToHost));
			unsigned long long q[DQ_CLASSES + 1];
			HIP_TRY(hipMemcpy(q, ws->counters + AC_DQ, sizeof q, hipMemcpyDeviceToHost));
			fprintf(stderr, "[kmahip] align: %llu tasks handed on to the general kernel (%d per wavefront round); %llu pending on queued problems: %llu wide, %llu mid, %llu narrow, %llu tiny (room for %lld each)\n",
			        c[0], A.per_round, q[DQ_CLASSES], q[0], q[3], q[1], q[2], (long long) A.dq_cap);
		}
		A.use_list = 0; A.per_round = 64;
	}
	if(ws->timing_on) {
		HIP_TRY(hipEventRecord(ev1, stream));
		if(!ws->events2) ws->events2 = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
		ws->events2->push_back({ev0, ev1});
	}
#ifdef KMAHIP_DIAG
	if(getenv("KMAHIP_DEBUG_TIMING")) {
		unsigned long long dbg[2];
		HIP_TRY(hipStreamSynchronize(stream));
		HIP_TRY(hipMemcpy(dbg, ws->counters + 13, sizeof dbg, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemset(ws->counters + 13, 0, sizeof dbg));
		fprintf(stderr, "[kmahip] align: slowest phase A %.1f us (task %llu), slowest phase B %.1f us (round of task %llu)\n",
		        (dbg[0] >> 32) / 100.0, dbg[0] & 0xFFFFFFFFull, (dbg[1] >> 32) / 100.0, dbg[1] & 0xFFFFFFFFull);
	}
#endif
	if(A.long_min) {
		int rcl = long_tasks(db, ws, reads, A, p, stream);
		if(rcl == KMAHIP_EOVERFLOW) {
			// a read with more MEMs than the pipeline has slots for: the lane kernel once more, for every task (it seeds the tasks
			// seed_tasks_kernel left out by itself, and reports its own lack of room as status 3: the caller raises mem_scale)
			A.long_min = 0;
			HIP_TRY(hipMemsetAsync(ws->counters + 7, 0, sizeof(unsigned long long), stream));
			hipLaunchKernelGGL((align_tasks_kernel<false, false>), agrid, dim3(ATHREADS), 0, stream, A);
			HIP_TRY(hipGetLastError());
			rcl = KMAHIP_OK;
		}
		if(rcl) return rcl;
	}
	ReduceArgs R;
	R.n_reads = n; R.rc_flag = cands->rc_flag; R.flag = cands->flag; R.T_off = cands->T_off; R.T = cands->T;
	R.t_score = A.t_score; R.t_alen = A.t_alen; R.t_start = A.t_start; R.t_end = A.t_end; R.t_norm = A.t_norm; R.t_tmpl = A.t_tmpl;
	R.k = (int) db->dev.kmersize; R.scoreT = p->scoreT;
	R.n_hits = out->n_hits; R.best_score = out->best_score; R.out_flag = out->flag;
	R.h_tmpl = out->tmpl; R.h_score = out->score; R.h_start = out->start; R.h_end = out->end;
	R.alignment_scores = (unsigned long long *) out->alignment_scores;
	R.uniq_alignment_scores = (unsigned long long *) out->uniq_alignment_scores;
	R.rec_mate = rec_mate; R.rec_rc = rec_rc; R.out_rc = out->rc; R.pe_mode = rec_mate != nullptr; R.PE = p->rw.PE; R.apm = ((p->apm >> 4) & 3) ? ((p->apm >> 4) & 3) - 1 : (p->apm & 3); R.pe_kind = pe_kind;
	R.priv = nullptr; R.priv_copies = 0; R.DB_size = db->info.DB_size; R.tasks_cap = tasks_cap;
	if(out->alignment_scores || out->uniq_alignment_scores) {
		const int64_t D = db->info.DB_size;
		int copies = (int) std::min<int64_t>(64, std::max<int64_t>(1, (256ll << 20) / (16 * D)));
		if(copies > 1) {
			if(ws->a_priv_cap < (int64_t) copies * 2 * D) {
				(void) hipFree(ws->a_priv);
				ws->a_priv = nullptr;
				HIP_TRY(hipMalloc((void **) &ws->a_priv, (size_t) copies * 2 * D * 8));
				ws->a_priv_cap = (int64_t) copies * 2 * D;
			}
			HIP_TRY(hipMemsetAsync(ws->a_priv, 0, (size_t) copies * 2 * D * 8, stream));
			R.priv = (unsigned long long *) ws->a_priv; R.priv_copies = copies;
		}
	}
	R.t_score_w = A.t_score; R.t_alen_w = A.t_alen; R.t_start_w = A.t_start; R.t_end_w = A.t_end; R.t_tmpl_w = A.t_tmpl;
	const unsigned rl_grid = 512;
	const bool lds_reduce = !R.pe_mode && R.DB_size <= RL_MAX_DB && (out->alignment_scores || out->uniq_alignment_scores) &&
	                        (n / rl_grid + RL_THREADS) * (int64_t) max_len < (1ll << 31);       // a workgroup's u32 sums cannot overflow
	if(lds_reduce) {
		ReduceArgs RL = R;
		RL.priv = nullptr; RL.priv_copies = 0;
		RL.alignment_scores = (unsigned long long *) out->alignment_scores; RL.uniq_alignment_scores = (unsigned long long *) out->uniq_alignment_scores;
		hipLaunchKernelGGL(reduce_reads_lds_kernel, dim3(rl_grid), dim3(RL_THREADS), 0, stream, RL);
		HIP_TRY(hipGetLastError());
		return KMAHIP_OK;
	}
	hipLaunchKernelGGL(reduce_reads_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, stream, R);
	if(R.priv_copies) hipLaunchKernelGGL(fold_scores_kernel, dim3((unsigned) ((2 * R.DB_size + 255) / 256)), dim3(256), 0, stream, R.priv, R.priv_copies,
	                                     R.DB_size, (unsigned long long *) out->alignment_scores, (unsigned long long *) out->uniq_alignment_scores);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

#ifdef KMAHIP_DIAG
// diagnostic build only: histogram of DP problems ([0..63] calls by q_len, [64..127] cells by q_len,
// [128..132] calls by mode k+2, [136..199] calls by t_len/4)
extern "C" int kmahip_diag_hist(unsigned long long *out256, int reset) {
	static unsigned long long *buf = nullptr;
	if(!buf) {
		if(hipMalloc((void **) &buf, 256 * 8) != hipSuccess) return -1;
		(void) hipMemset(buf, 0, 256 * 8);
		(void) hipMemcpyToSymbol(HIP_SYMBOL(g_diag_hist), &buf, sizeof buf);
	}
	(void) hipDeviceSynchronize();
	if(out256) (void) hipMemcpy(out256, buf, 256 * 8, hipMemcpyDeviceToHost);
	if(reset) (void) hipMemset(buf, 0, 256 * 8);
	return 0;
}
#endif

int kmahip_launch_align_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                           const kmahip_params *p, kmahip_hits *out, hipStream_t stream) {
	return launch_align(db, ws, reads, cands, nullptr, nullptr, nullptr, p, out, stream);
}

int kmahip_launch_align_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                           const kmahip_params *p, kmahip_hits *out, int32_t *pe_kind, hipStream_t stream) {
	if(!recs || !pe_kind || (reads->n_reads & 1)) { kmahip_set_error("paired align needs interleaved mates and a kind array"); return KMAHIP_EINVAL; }
	kmahip_cands c;
	c.rc_flag = recs->rc_flag; c.flag = recs->flag; c.T_off = recs->R_off; c.T = recs->T; c.T_cap = recs->T_cap;
	return launch_align(db, ws, reads, &c, recs->mate, recs->rc, pe_kind, p, out, stream);
}

// stage 3c launcher: scratch lives in the workspace, sized by the longest read
// the scratch of the lane-per-read traceback, sized by the longest read: made (or made anew) here, so that a caller that knows what
// is coming can have it made while something else is going on (kmahip_trace_reserve: the batched session does, beside stage 1)
struct TraceGeom { int mem_cap, ncols, ops_cap; int64_t e_cap, lanes; };
static int trace_reserve(kmahip_ws *ws, int max_len, int64_t n, TraceGeom &G) {
	G.mem_cap = (max_len <= 1024 ? 64 : max_len / 8) * std::max(1, ws->mem_scale);
	G.ncols = max_len + 72;
	G.ops_cap = 2 * max_len + 256;
	// move matrix per lane: a banded tail (band 64 + 64) or a full join whose shorter side is within the band
	G.e_cap = std::max<int64_t>((int64_t) 134 * (max_len + 68), (int64_t) (max_len / 2 + 4) * (max_len + 4));
	G.e_cap = std::min<int64_t>(G.e_cap, 16ll << 20);
	// one lane per read in flight; the kernel is bound by dependent loads, so it wants every wave slot of the chip (4 per SIMD
	// at its register count = 262 144 lanes): 10 GB of scratch for 150-base reads, shrunk for longer ones to stay within 24 GB
	int64_t lanes = 262144;
	const int64_t per_lane = G.e_cap + (int64_t) (7 * (G.mem_cap + 1) + 4 * G.ncols + G.ops_cap) * 4;
	while(lanes > 256 && lanes * per_lane > (24ll << 30)) lanes >>= 1;
	lanes = std::min<int64_t>(lanes, ((n + 255) / 256) * 256);
	G.lanes = lanes;
	if(ws->t_lanes != lanes || ws->t_max_len != max_len || ws->t_mem_cap != G.mem_cap) {
		(void) hipFree(ws->t_s32); (void) hipFree(ws->t_E);
		ws->t_s32 = nullptr; ws->t_E = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->t_s32, (size_t) lanes * (7 * (G.mem_cap + 1) + 4 * G.ncols + G.ops_cap) * 4));
		HIP_TRY(hipMalloc((void **) &ws->t_E, (size_t) lanes * G.e_cap));
		ws->t_lanes = lanes; ws->t_max_len = max_len; ws->t_mem_cap = G.mem_cap;
	}
	return KMAHIP_OK;
}
int kmahip_trace_reserve(kmahip_ws *ws, int max_len, int64_t n) {
	if(max_len <= 0 || max_len > 1024 || n <= 0) return KMAHIP_OK;          // (longer reads take the pipeline of longtrace.hip)
	TraceGeom G;
	return trace_reserve(ws, max_len, n, G);
}

int kmahip_launch_trace(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                        const uint8_t *tmpl_ok, const kmahip_params *p, kmahip_traces *out, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	if(n < 0 || !flag || !tmpl || !p || !out || !out->stats || !out->ops_off || !out->n_ops || (out->ops_cap > 0 && !out->ops)) { kmahip_set_error("bad arguments"); return KMAHIP_EINVAL; }
	if(!db->dev.tpos_slots) { kmahip_set_error("index has no .length.b/.seq.b: stage 3c unavailable"); return KMAHIP_EINVAL; }
	if(n == 0) return KMAHIP_OK;
	const int max_len = reads->max_len;
	if(max_len <= 0 || max_len > (1 << 20)) { kmahip_set_error("kmahip_reads.max_len must be set (<= 2^20) for the trace stage"); return KMAHIP_EINVAL; }
	// the pipeline of longtrace.hip (a wavefront per read, DP problems batched by size) instead of one lane per read with its move
	// matrix in HBM: for reads that carry hundreds of MEMs, or everything with KMAHIP_TRACE=pipeline (=lanes forces this kernel)
	{
		const char *mode = getenv("KMAHIP_TRACE");
		const bool pipeline = mode ? !strcmp(mode, "pipeline") : max_len > 1024;
		if(pipeline) return kmahip_launch_longtrace(db, ws, reads, tmpl, 0, flag, tmpl_ok, 0, p, out, nullptr, stream);
	}
	TraceGeom G;
	{
		const int rc = trace_reserve(ws, max_len, n, G);
		if(rc) return rc;
	}
	const int mem_cap = G.mem_cap, ncols = G.ncols, ops_cap = G.ops_cap;
	const int64_t e_cap = G.e_cap, lanes = G.lanes;
	if(!ws->counters) { HIP_TRY(hipMalloc((void **) &ws->counters, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); HIP_TRY(hipMemset(ws->counters, 0, KMAHIP_N_COUNTERS * sizeof(unsigned long long))); }
	HIP_TRY(hipMemsetAsync(ws->counters, 0, sizeof(unsigned long long), stream));
	HIP_TRY(hipMemsetAsync(ws->counters + 3, 0, 2 * sizeof(unsigned long long), stream));
	if(ws->t_queue_cap < n) {
		(void) hipFree(ws->t_queue);
		ws->t_queue = nullptr; ws->t_queue_cap = 0;
		HIP_TRY(hipMalloc((void **) &ws->t_queue, (size_t) n * 2 * sizeof(int32_t)));
		ws->t_queue_cap = n;
	}
	TraceArgs A;
	A.db = db->dev; A.n_reads = n; A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.flag = flag; A.tmpl = tmpl; A.tmpl_ok = tmpl_ok;
	A.q_start = reads->q_start; A.q_end = reads->q_end;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1; A.Wl = p->rw.Wl;
	for(int i = 0; i < 25; ++i) A.d[i] = p->rw.d[i / 5][i % 5];
	A.minlen = p->minlen; A.mq = p->mq; A.scoreT = p->scoreT; A.mrc = p->mrc; A.ts = p->ts;
	A.s32 = ws->t_s32; A.rows = ws->t_s32 + (size_t) lanes * 7 * (mem_cap + 1);
	A.ops_s = (uint32_t *) (A.rows + (size_t) lanes * 4 * ncols);
	A.E = ws->t_E; A.lanes = lanes; A.e_cap = e_cap; A.mem_cap = mem_cap; A.ncols = ncols; A.ops_cap = ops_cap;
	A.o_stats = out->stats; A.o_off = out->ops_off; A.o_nops = out->n_ops; A.ops = out->ops; A.ops_pool_cap = out->ops_cap;
	A.counters = ws->counters;
	// pass 1 settles every read whose DP problems are provably ungapped (diag_proof above) and lists the others; pass 2 runs
	// the listed ones with their move matrices. KMAHIP_TRACE=lanes1: everything in one pass, as before.
	const int M = p->rw.M, MM = p->rw.MM, W1 = p->rw.W1, U = p->rw.U;
	bool plain = M > 0 && MM < 0 && W1 < 0 && U < 0 && MM > W1 && MM - M > 2 * W1 && MM - M > W1 + U;
	for(int i = 0; i < 4 && plain; ++i) for(int j = 0; j < 4; ++j) plain = plain && p->rw.d[i][j] == (i == j ? M : MM);
	{
		const char *mode = getenv("KMAHIP_TRACE");
		if(mode && !strcmp(mode, "lanes1")) plain = false;
	}
	A.q_in = nullptr; A.q_out = ws->t_queue; A.q_in_cnt = 3; A.q_out_cnt = 3; A.fast = plain ? 1 : 0;
	A.gap_m_max = plain ? (M - 2 * W1 - 1) / (M - MM) : 0;
	A.lds_bytes = 0; A.lds_ncols = 0;
	if(plain) {
		// the first pass holds no move matrix and no DP rows: it fits five waves per SIMD where the full kernel fits four, and its
		// lanes lay their MEM arrays and run slots over the room of the rows (same allocation, more lanes)
		TraceArgs F = A;
		int64_t lanes_fast = lanes / 4 * 5;
		while(lanes_fast > lanes && lanes_fast * (int64_t) (7 * (mem_cap + 1) + ops_cap) > lanes * (int64_t) (7 * (mem_cap + 1) + 4 * ncols + ops_cap)) lanes_fast -= 256;
		lanes_fast = std::min<int64_t>(lanes_fast, ((n + 255) / 256) * 256);
		F.lanes = lanes_fast;
		F.ops_s = (uint32_t *) (ws->t_s32 + (size_t) lanes_fast * 7 * (mem_cap + 1));
		hipLaunchKernelGGL(trace_kernel<true>, dim3((unsigned) (lanes_fast / 256)), dim3(256), 0, stream, F);
	} else hipLaunchKernelGGL(trace_kernel<false>, dim3((unsigned) (lanes / 256)), dim3(256), 0, stream, A);
	HIP_TRY(hipGetLastError());
	if(plain) {
		// pass 2: the reads put off, with their move matrices in HBM. (KMAHIP_TRACE_LDS=1 puts a pass with the matrices of small
		// problems in LDS in between -- 1008 bytes per lane: rows of 24 columns + 624 move bytes; measured on the 10 M-read
		// workload it takes a third of the put-off reads and the kernel time stays that of the slowest lane of the last pass:
		// 27.9 ms for the three passes against 23.1 for two. Kept for workloads whose put-off problems are all small.)
		const bool lds_pass = getenv("KMAHIP_TRACE_LDS") && atoi(getenv("KMAHIP_TRACE_LDS"));
		A.q_in = ws->t_queue; A.q_out = ws->t_queue + n; A.q_in_cnt = 3; A.q_out_cnt = 4; A.fast = 0;
		if(lds_pass) {
			A.lds_bytes = 1008; A.lds_ncols = 24;
			const unsigned g2 = (unsigned) std::min<int64_t>(lanes / 64, 256 * 2 * 8);
			hipLaunchKernelGGL(trace_kernel<false>, dim3(g2), dim3(64), 64 * 1008, stream, A);
			HIP_TRY(hipGetLastError());
			A.q_in = ws->t_queue + n; A.q_in_cnt = 4;
		}
		A.q_out = nullptr; A.lds_bytes = 0; A.lds_ncols = 0;
		hipLaunchKernelGGL(trace_kernel<false>, dim3((unsigned) (lanes / 256)), dim3(256), 0, stream, A);
		HIP_TRY(hipGetLastError());
		if(getenv("KMAHIP_DEBUG_TIMING")) {
			unsigned long long put_off[2] = {0, 0};
			HIP_TRY(hipStreamSynchronize(stream));
			HIP_TRY(hipMemcpy(put_off, ws->counters + 3, sizeof put_off, hipMemcpyDeviceToHost));
			if(lds_pass) fprintf(stderr, "[kmahip] trace: %lld reads, %llu put off to the pass with matrices in LDS, %llu of them to the pass with matrices in HBM\n",
			                     (long long) n, put_off[0], put_off[1]);
			else fprintf(stderr, "[kmahip] trace: %lld reads, %llu put off to the pass with move matrices\n", (long long) n, put_off[0]);
		}
	}
	return KMAHIP_OK;
}
