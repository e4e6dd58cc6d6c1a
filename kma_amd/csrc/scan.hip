// scan.hip -- stage 2 of KMA on gfx950: k-mer extraction, probe of the template
// k-mer table in HBM and per-read candidate-template scoring for `-1t1`
// single-end reads and `-apm p` pairs.  Behaviour restated from save_kmers
// (savekmers.c:2442-3065), get_kmers_for_pair / save_kmers_penaltyPair
// (:427-688, :3572-3777), getBestMatch (:273-294) and hashMap_getGlobal
// (hashmapkma.c:149-178); the structure is new:
//
//   work item  = one (read, strand)
//   prefilter  = scan_prefilter_kernel: one lane per item probes every k-th k-mer
//                (:2477-2495; presence bits in L2 for small databases); survivors go
//                to a device-wide list, one global atomic per 512 items
//   scan       = scan_se_kernel: 16 live items x 16 lanes per workgroup. Phase 1: each
//                lane anchors its 9 k-mer starts with one hash probe and walks along the
//                concatenated templates (`cat` / `vs_id`); every run of equal value
//                lists ORs its positions into the mask of that LIST (16-slot table per
//                item in LDS). Phase 2a: one thread per (item, list) reads the list once
//                and ORs the mask into each listed TEMPLATE's mask. Phase 2b: one thread
//                per (item, template) folds its mask into the score (what the reference's
//                run-length machine :2511-2706 computes, per template instead of per
//                position). Finish: best score + tied templates in first-seen order.
//   overflow   = items with more than 14 candidates: scan_dense_kernel, the reference's
//                own sequential formulation on DB_size-wide tables in HBM, one wavefront
//                per item
//   combine    = per read strand pick / tie merge (:3037-3062), CSR output; for pairs
//                pair_penalty_kernel (getFirstPen / getSecondBestPen / getF_Best)
//
// The words of a read are staged in LDS in STRAND orientation, so no k-mer or walk
// window is reverse-complemented after staging.
#include "kmahip_internal.h"
#include <cstdlib>
#include <climits>

namespace {


constexpr int GROUP = 16;             // active items scored together
constexpr int THREADS = 256;
// (tools/scan_shape_sweep.sh, round 4, scan_se_kernel per 10 M reads: 256 threads / 7 waves per SIMD 8.74 ms, 64 / 7 8.40, 64 / 8 8.30,
// 128 / 7 8.33, 128 / 8 8.11 -- a barrier waits for fewer wavefronts, and the smaller tables let eight waves live on a SIMD)
#ifndef KMAHIP_STHREADS
#define KMAHIP_STHREADS 128
#endif
#ifndef KMAHIP_SCAN_WAVES
#define KMAHIP_SCAN_WAVES 8
#endif
constexpr int STHREADS = KMAHIP_STHREADS;   // threads of one scan workgroup: SG items x 16 lanes
constexpr int SG = STHREADS / 16;           // strand items a scan workgroup scores together (16 at 256 threads; 4 = a workgroup of ONE wavefront, whose barriers cost nothing)
constexpr int CHUNK = 136;            // k-mer start positions per pass
constexpr int MW = 5;                 // hit-mask words per candidate (>= CHUNK / 32)
constexpr int SW = 7;                 // staged u64 words per item and pass
// hashed candidate slots per item in LDS (template parameter TS of the scan kernel; at most TS - 2 occupied): 16 in the
// first tier (8 waves / SIMD), 64 in the second tier that re-does the items whose candidates did not fit -- a 50 k-gene
// database with ten variants per family puts twenty and more templates on half of the reads (one chance k-mer hit in
// another family brings all its variants), and the wavefront-per-item overflow kernel is 30x slower per item
constexpr int TS1 = 16, TS2 = 64;
constexpr unsigned TIER2_GRID = 256 * 3 * (256 / STHREADS);        // second tier: three workgroups (of 256 threads) per CU fit its LDS
__host__ __device__ constexpr int ilog2c(int x) { return x <= 1 ? 0 : 1 + ilog2c(x >> 1); }
constexpr uint32_t T_EMPTY = 0xFFFFFFFFu;
constexpr int VSLOTS = 16;            // hashed value-list slots per item in LDS (distinct lists seen in one pass); a list
                                      // that finds all of them taken is expanded directly
constexpr int QCAP = 64 * SG;         // entries of a workgroup's queue of left-over k-mer starts (a fuller one is finished by a rescan)
constexpr int INL = 2;                // inline result slots per strand item (no allocation round trip for the usual 1-2 ties)
constexpr uint32_t MISS = 0xFFFFFFFFu;
constexpr uint32_t NONE = 0xFFFFFFFEu;

struct ScanArgs {
	DevDB db;
	int64_t n_reads;
	const uint64_t *seq;
	const int64_t *seq_off;
	const int32_t *len;
	const int32_t *N;
	const int64_t *N_off;
	int M, MM, U, W1, exhaustive;
	int ablate;      // diagnostic builds only (KMAHIP_DIAG): 1 skip score machines, 2 skip phase-1 probes, 4 skip prefilter probes
	int32_t *item_score;
	int32_t *item_n;
	int64_t *item_off;
	int32_t *pool;
	int64_t pool_cap;
	unsigned long long *counters;
	int64_t *overflow_items;
	// what a launch of scan_se_kernel / scan_dense_kernel reads its items from and where it files what does not fit:
	// in_items[0 .. counters[in_count]), out_over[counters[out_count]++]
	const int64_t *in_items;
	int64_t *out_over;
	int in_count, out_count;
	int32_t *dense;
	int64_t dense_slots;
	int64_t *active_items; // strand items that passed the prefilter, in no particular order (counters[C_NACT] of them)
	int64_t pool_tail0;  // pool[0 .. pool_tail0) = INL inline slots per strand item; longer lists are bump-allocated after it
	int mode;            // 0: best templates per strand (save_kmers); 1: every candidate + score + hit count (get_kmers_for_pair)
	int32_t *pool_sc;    // mode 1: scores parallel to pool
	int64_t cat_bases;   // > 0: the prefilter files the template diagonal of its hit with every live strand item (bits 33.. of the
	                     // list entry: position in `cat` minus strand position of the hit k-mer, + DIAG_BIAS; DIAG_NONE: none) and the
	                     // scan's lanes try that diagonal before they probe
};

constexpr int64_t ITEM_MASK = (1ll << 33) - 1;          // a strand item: (read << 1) | strand, reads < 2^31
constexpr int64_t DIAG_BIAS = 1ll << 20, DIAG_NONE = 0x7FFFFFFFll;      // (reads are at most 2^20 bases, `cat` below 2^30 when diagonals are filed)

enum { C_POOL = 0, C_STATUS = 1, C_NOVER = 2, C_PROBES = 3, C_VALS = 4, C_ACTIVE = 5, C_HASH = 6, C_PPOOL = 7, C_NACT = 8, C_PREF = 9, C_NOVER2 = 10, N_COUNTERS = KMAHIP_N_COUNTERS };

// two probes whose home buckets travel together (used after a miss: the k-mer starts behind a mismatch miss in a row)
__device__ __forceinline__ void probe2(const DevDB &db, uint32_t key1, uint32_t key2, uint32_t &r1, uint32_t &r2) {
	const uint32_t sh = 32u - db.nb_log2;
	const uint32_t nbm = (1u << db.nb_log2) - 1u;
	uint32_t b1 = (key1 * 0x9E3779B1u) >> sh, b2 = (key2 * 0x9E3779B1u) >> sh;
	const uint4 *p1 = reinterpret_cast<const uint4 *>(db.slots + (size_t) b1 * KMAHIP_BUCKET_SLOTS);
	const uint4 *p2 = reinterpret_cast<const uint4 *>(db.slots + (size_t) b2 * KMAHIP_BUCKET_SLOTS);
	uint4 a1 = p1[0], c1 = p1[1], a2 = p2[0], c2 = p2[1];
	for(;;) {
		if(a1.x == key1 && a1.y != KMAHIP_EMPTY_VI) { r1 = a1.y; break; }
		if(a1.z == key1 && a1.w != KMAHIP_EMPTY_VI) { r1 = a1.w; break; }
		if(c1.x == key1 && c1.y != KMAHIP_EMPTY_VI) { r1 = c1.y; break; }
		if(c1.z == key1 && c1.w != KMAHIP_EMPTY_VI) { r1 = c1.w; break; }
		if(c1.w == KMAHIP_EMPTY_VI) { r1 = 0xFFFFFFFFu; break; }
		b1 = (b1 + 1u) & nbm;
		p1 = reinterpret_cast<const uint4 *>(db.slots + (size_t) b1 * KMAHIP_BUCKET_SLOTS);
		a1 = p1[0]; c1 = p1[1];
	}
	for(;;) {
		if(a2.x == key2 && a2.y != KMAHIP_EMPTY_VI) { r2 = a2.y; break; }
		if(a2.z == key2 && a2.w != KMAHIP_EMPTY_VI) { r2 = a2.w; break; }
		if(c2.x == key2 && c2.y != KMAHIP_EMPTY_VI) { r2 = c2.y; break; }
		if(c2.z == key2 && c2.w != KMAHIP_EMPTY_VI) { r2 = c2.w; break; }
		if(c2.w == KMAHIP_EMPTY_VI) { r2 = 0xFFFFFFFFu; break; }
		b2 = (b2 + 1u) & nbm;
		p2 = reinterpret_cast<const uint4 *>(db.slots + (size_t) b2 * KMAHIP_BUCKET_SLOTS);
		a2 = p2[0]; c2 = p2[1];
	}
}

__device__ __forceinline__ uint32_t probe(const DevDB &db, uint32_t key) {
	const uint32_t sh = 32u - db.nb_log2;
	const uint32_t nbm = (1u << db.nb_log2) - 1u;
	uint32_t b = (key * 0x9E3779B1u) >> sh;
	for(;;) {
		const uint4 *p = reinterpret_cast<const uint4 *>(db.slots + (size_t) b * KMAHIP_BUCKET_SLOTS);
		const uint4 a = p[0], c = p[1];
		if(a.x == key && a.y != KMAHIP_EMPTY_VI) return a.y;
		if(a.z == key && a.w != KMAHIP_EMPTY_VI) return a.w;
		if(c.x == key && c.y != KMAHIP_EMPTY_VI) return c.y;
		if(c.z == key && c.w != KMAHIP_EMPTY_VI) return c.w;
		if(c.w == KMAHIP_EMPTY_VI) return MISS; // bucket not full: the key cannot be further on
		b = (b + 1u) & nbm;
	}
}

// reverse complement of a k-mer held in the low 2k bits
__device__ __forceinline__ uint64_t revcomp_kmer(uint64_t x, int k) {
	x = ~x;
	x = __brevll(x);
	x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
	return x >> (64 - 2 * k);
}

__device__ __forceinline__ uint64_t kmer_from(uint64_t lo, uint64_t hi, int q, int k) {
	const int ip = (q & 31) << 1;
	const uint64_t x = (lo << ip) | ((hi >> 1) >> (63 - ip));      // branch-free: nothing of hi when ip == 0
	return x >> (64 - 2 * k);
}

// 32 bases starting at base `pos` of a 2-bit word array (MSB first). Always reads word and word+1 (every array carries a
// pad word), as ONE 16-byte access and without a branch: a conditional second load would wait for the first one.
__device__ __forceinline__ uint64_t win2(const uint64_t *w, int64_t pos) {
	const int ip = (int) (pos & 31) << 1;
	const uint64_t *p = w + (pos >> 5);
	const uint64_t w0 = p[0], w1 = p[1];
	return (w0 << ip) | ((w1 >> 1) >> (63 - ip));
}

// 32 bases of the read in STRAND orientation starting at strand position i (strand 1 = reverse complement of
// the forward words `w` of a read of length L); bases past the read end are garbage
__device__ __forceinline__ uint64_t strand_win(const uint64_t *w, int L, int strand, int i) {
	if(!strand) return win2(w, i);
	const int s0 = L - 32 - i;
	uint64_t x = (s0 >= 0) ? win2(w, s0) : (w[0] >> ((-s0) << 1));
	x = __brevll(~x);
	return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// cost of bridging `gaps` missed k-mer starts between two hits of a template.
// savekmers.c:2522-2569 (run form) and :2590-2627 (per-template form) coincide
// when mlen == kmersize (enforced at kmahip_db_open), on both strands.
__device__ __forceinline__ int bridge(int gaps, int k, int M, int MM, int U, int W1) {
	if(gaps == 0) return M;
	if(gaps == k) return k * M + MM;
	if(k < gaps) {
		int g = gaps - (k - 1), mm, m;
		if(g <= 2) { mm = g; m = 0; }
		else {
			mm = g / k + (g % k ? 1 : 0);
			mm = max(2, mm);
			m = min(min(g - mm, k), mm);
		}
		const int sub = mm * MM + m * M, ind = W1 + (g - 1) * U;
		return k * M + (ind <= sub ? sub : ind);
	}
	return gaps * M + (k - gaps) * U + W1;
}

__device__ __forceinline__ uint32_t value_at(const DevDB &db, uint32_t vi, int i) {
	return db.values_u16 ? (uint32_t) db.values16[vi + i] : db.values32[vi + i];
}

// does the forward window [q, q+k) hold an N ?  Nl = sorted forward N positions
__device__ __forceinline__ bool window_has_N(const int32_t *Nl, int nN, int q, int k) {
	int lo = 0, hi = nN;
	while(lo < hi) { const int mid = (lo + hi) >> 1; if(Nl[mid] < q) lo = mid + 1; else hi = mid; }
	return lo < nN && Nl[lo] < q + k;
}

// i-th (1-based) N position in strand coordinates; i == nN + 1 -> seqlen sentinel
__device__ __forceinline__ int n_strand(const int32_t *Nl, int nN, int L, int strand, int i) {
	if(i > nN) return L;
	return strand ? (L - 1 - Nl[nN - i]) : Nl[i - 1];
}

// first set bit at or after `from` in an MW-word mask laid out with stride `st`; MW*32 if none
__device__ __forceinline__ int mask_next(const uint32_t *m, int st, int from, bool want_set) {
	int w = from >> 5;
	if(w >= MW) return MW * 32;
	uint32_t x = want_set ? m[w * st] : ~m[w * st];
	x &= 0xFFFFFFFFu << (from & 31);
	for(;;) {
		if(x) return (w << 5) + __ffs((int) x) - 1;
		if(++w >= MW) return MW * 32;
		x = want_set ? m[w * st] : ~m[w * st];
	}
}

// claim / find template t's slot in item g's candidate table; -1 when the table is full
template <int TSLOTS>
__device__ __forceinline__ int template_slot(uint32_t t, int g, uint32_t *t_id, int32_t *t_cnt) {
	constexpr int TMAX = TSLOTS - 2;
	const uint32_t h = (t * 0x9E3779B1u) >> (32 - ilog2c(TSLOTS));
	// rolled on purpose: unrolled, the 16 probes nest 16 exec masks (SGPR spills) and the body is inlined many times over
#pragma unroll 1
	for(int x = 0; x < TSLOTS; ++x) {
		const int sidx = (int) ((h + x) & (TSLOTS - 1)) * SG + g;
		const uint32_t old = atomicCAS(&t_id[sidx], T_EMPTY, t);
		if(old == T_EMPTY) {
			if(atomicAdd(&t_cnt[g], 1) >= TMAX) return -1;
			return sidx;
		}
		if(old == t) return sidx;
	}
	return -1;
}

// rare path: OR positions [rs, re) of the pass into the hit mask of every template of list vi
template <int TSLOTS>
__device__ __forceinline__ void expand_list(const DevDB &db, uint32_t vi, int rs, int re, int g,
                                         uint32_t *t_id, int32_t *t_cnt, uint32_t *t_mask, int32_t *s_over) {
	const uint32_t cnt = value_at(db, vi, 0);
	for(uint32_t i = 1; i <= cnt; ++i) {
		const int slot = template_slot<TSLOTS>(value_at(db, vi, (int) i), g, t_id, t_cnt);
		if(slot < 0) { s_over[g] = 1; return; }
		for(int w = rs >> 5; w <= (re - 1) >> 5; ++w) {
			const int lo = max(rs, w << 5) & 31, hi = min(re, (w + 1) << 5) - (w << 5);   // bits [lo, hi)
			const uint32_t m = (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & (0xFFFFFFFFu << lo);
			atomicOr(&t_mask[w * TSLOTS * SG + slot], m);
		}
	}
}

// ---- prefilter (savekmers.c:2477-2495): every k-th k-mer of every N-free segment of both strands is probed by
// 8 adjacent lanes per strand item; a strand with no hit is finished (no candidates), the others are appended to
// the device-wide list of active items, so that the scan kernel below only ever sees full groups of live items.
// One workgroup filters PF_ITEMS items and appends its survivors with a single global atomic.
constexpr int PF_PLANES = 1;                               // lanes sharing one item's probes (8: 0.71 ms, 2: 0.51, 1 with paired probes: see DESIGN)
constexpr int PF_BLOCK = THREADS / PF_PLANES;              // items per round
constexpr int PF_ROUNDS = 512 / PF_BLOCK;
constexpr int PF_ITEMS = PF_BLOCK * PF_ROUNDS;             // items per workgroup

template <bool STATS>
__global__ __launch_bounds__(THREADS) void scan_prefilter_kernel(const ScanArgs A) {
	__shared__ int64_t s_list[PF_ITEMS];
	__shared__ uint32_t s_n, s_np, s_nt;
	__shared__ unsigned long long s_base;
	const DevDB &db = A.db;
	const int tid = threadIdx.x;
	const int k = (int) db.kmersize;
	const int plane = tid & (PF_PLANES - 1);
	if(tid == 0) { s_n = 0; s_np = 0; s_nt = 0; }
	__syncthreads();
	uint32_t nprobe = 0, ntable = 0;      // k-mers resolved / of them by a gather into the probe table
	for(int rd = 0; rd < PF_ROUNDS; ++rd) {
		const int64_t item = (int64_t) blockIdx.x * PF_ITEMS + rd * PF_BLOCK + (tid / PF_PLANES);
		const int64_t r = item >> 1;
		bool hit = false;
		int64_t diag = DIAG_NONE;          // template diagonal of the hit: its position in `cat` minus its strand position in the read
		if(r < A.n_reads) {
			const int L = A.len[r], strand = (int) (item & 1), npos = L - k + 1;
			if(npos > 0) {
				const uint64_t *rs = A.seq + A.seq_off[r];
				const int64_t no = A.N_off[r];
				const int nN = (int) (A.N_off[r + 1] - no);
#ifdef KMAHIP_DIAG
				if(A.exhaustive || (A.ablate & 4)) {
#else
				if(A.exhaustive) {
#endif
					hit = strand == 0 || A.exhaustive;
				} else if(nN == 0 && db.kbits) {
					// small database: the presence bits of the stride k-mers are fetched first (L2-resident, nine in flight), and only
					// those that pass -- the hit of the right strand, the ~13 % false positives of the wrong one -- are probed for real
					const int nst = (npos + k - 1) / k;
					auto stride_kmer = [&](int st) -> uint32_t {
						const int q = strand ? (L - k - st * k) : st * k;
						uint64_t km = kmer_from(rs[q >> 5], rs[(q >> 5) + 1], q, k);
						if(strand) km = revcomp_kmer(km, k);
						return (uint32_t) km;
					};
					int first_hit = -1;
					for(int sb = 0; sb < nst; sb += 9) {
						uint32_t kms[9], wd[9];
#pragma unroll
						for(int u = 0; u < 9; ++u) {
							kms[u] = 0; wd[u] = 0;
							if(sb + u < nst) {
								kms[u] = stride_kmer(sb + u);
								wd[u] = db.kbits[((kms[u] * KMAHIP_KBITS_MUL) >> db.kbits_shift) >> 5];
							}
						}
#pragma unroll
						for(int u = 0; u < 9; ++u) {
							if(first_hit >= 0 || sb + u >= nst) continue;
							++nprobe;
							const uint32_t h = (kms[u] * KMAHIP_KBITS_MUL) >> db.kbits_shift;
							if(!((wd[u] >> (h & 31)) & 1u)) continue;
							++ntable;
							const uint32_t gp = probe(db, kms[u]);
							if(gp != MISS) { first_hit = sb + u; diag = (int64_t) gp - (int64_t) (sb + u) * k; }
						}
						if(first_hit >= 0) break;
					}
					hit = first_hit >= 0;
				} else if(nN == 0) {
					// two stride positions per step, their home buckets in flight together
					for(int j = plane * k; j < npos; j += 2 * PF_PLANES * k) {
						const int q = strand ? (L - k - j) : j;
						uint64_t km = kmer_from(rs[q >> 5], rs[(q >> 5) + 1], q, k);
						if(strand) km = revcomp_kmer(km, k);
						++nprobe; ++ntable;
						const int j2 = j + PF_PLANES * k;
						if(j2 < npos) {
							const int q2 = strand ? (L - k - j2) : j2;
							uint64_t km2 = kmer_from(rs[q2 >> 5], rs[(q2 >> 5) + 1], q2, k);
							if(strand) km2 = revcomp_kmer(km2, k);
							uint32_t r1, r2;
							probe2(db, (uint32_t) km, (uint32_t) km2, r1, r2);
							if(r1 != MISS) { hit = true; diag = (int64_t) r1 - j; break; }
							++nprobe; ++ntable;
							if(r2 != MISS) { hit = true; diag = (int64_t) r2 - j2; break; }
						} else { const uint32_t r1 = probe(db, (uint32_t) km); if(r1 != MISS) { hit = true; diag = (int64_t) r1 - j; break; } }
					}
				} else if(plane == 0) {
					// rare: walk the N-free segments exactly like savekmers.c:2483-2495
					const int32_t *Nl = A.N + no;
					int j = 0;
					for(int i = 1; i <= nN + 1 && !hit; ++i) {
						const int segend = n_strand(Nl, nN, L, strand, i);
						for(; j < segend - k + 1 && !hit; j += k) {
							const int q = strand ? (L - k - j) : j;
							const int w = q >> 5;
							uint64_t km = kmer_from(rs[w], rs[w + 1], q, k);
							if(strand) km = revcomp_kmer(km, k);
							++nprobe; ++ntable;
							if(probe(db, (uint32_t) km) != MISS) hit = true;
						}
						j = segend + 1;
					}
				}
			}
		}
		// OR over the 8 lanes of the item (they sit next to each other in one wavefront)
		const unsigned long long bal = __ballot(hit);
		const bool any = ((bal >> ((tid & 63) & ~(PF_PLANES - 1))) & ((1ull << PF_PLANES) - 1ull)) != 0ull;
		if(plane == 0 && r < A.n_reads) {
			if(any) s_list[atomicAdd(&s_n, 1u)] = item | ((A.cat_bases > 0 && diag != DIAG_NONE ? diag + DIAG_BIAS : DIAG_NONE) << 33);
			else { A.item_score[item] = 0; A.item_n[item] = 0; A.item_off[item] = 0; }
		}
	}
	if(STATS && nprobe) { atomicAdd(&s_np, nprobe); atomicAdd(&s_nt, ntable); }
	__syncthreads();
	const uint32_t nact = s_n;
	if(tid == 0) {
		s_base = nact ? atomicAdd(&A.counters[C_NACT], (unsigned long long) nact) : 0ull;
		if(STATS) {
			atomicAdd(&A.counters[C_PROBES], (unsigned long long) s_np);
			atomicAdd(&A.counters[C_HASH], (unsigned long long) s_nt);
			atomicAdd(&A.counters[C_PREF], (unsigned long long) s_np);
			atomicAdd(&A.counters[C_ACTIVE], (unsigned long long) nact);
		}
	}
	__syncthreads();
	for(uint32_t i = tid; i < nact; i += THREADS) A.active_items[s_base + i] = s_list[i];
}

// ---- scan: one workgroup = GROUP active strand items, 16 lanes each ----------------------------------------
// Workgroup-wide compaction of the threads whose `flag` is set: afterwards list[0 .. total) holds their thread ids, so the
// first `total` threads can take one occupied table slot each -- the hashed tables are a third full, and a wavefront costs
// the same whether 20 or 64 of its lanes have work. Two barriers inside.
__device__ __forceinline__ int compact_threads(bool flag, int tid, int32_t *wcnt, uint16_t *list) {
	const unsigned long long m = __ballot(flag);
	const int wave = tid >> 6, lane = tid & 63;
	if(lane == 0) wcnt[wave] = __popcll(m);
	__syncthreads();
	int before = 0, total = 0;
#pragma unroll
	for(int w = 0; w < STHREADS / 64; ++w) { const int c = wcnt[w]; if(w < wave) before += c; total += c; }
	if(flag) list[before + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t) tid;
	__syncthreads();
	return total;
}

template <bool STATS, int MODE, int TSLOTS>
__global__ __launch_bounds__(STHREADS, TSLOTS == TS1 ? KMAHIP_SCAN_WAVES : (STHREADS == 256 ? 2 : 3)) void scan_se_kernel(const ScanArgs A) {
	__shared__ uint32_t v_id[VSLOTS * SG];              // value-list offset per slot (MISS = free)
	__shared__ uint32_t v_mask[MW * VSLOTS * SG];       // positions of the pass whose k-mer carries that value list
	// forward words: if every read of the group fits in SW-1 words they are staged ONCE and serve all passes;
	// otherwise (s_anylong) they are staged per pass
	__shared__ uint64_t w_lds[SG * SW];
	__shared__ int32_t s_anylong;
	__shared__ uint32_t t_id[TSLOTS * SG];
	__shared__ uint32_t t_mask[MW * TSLOTS * SG];
	__shared__ int32_t t_score[TSLOTS * SG];
	__shared__ int32_t t_last[TSLOTS * SG];
	__shared__ int32_t t_first[TSLOTS * SG];
	__shared__ int32_t t_cnt[SG], s_over[SG], s_hits[SG], s_best[SG], s_nb[SG];
	__shared__ int64_t s_off[SG];
	__shared__ int32_t s_len[SG], s_nN[SG];
	__shared__ int64_t s_soff[SG], s_noff[SG], s_item[SG];
	__shared__ int32_t s_diag[SG];  // the prefilter's diagonal (+ DIAG_BIAS), DIAG_NONE: none
	__shared__ int32_t s_gmax;
	__shared__ int32_t s_wcnt[STHREADS / 64];
	__shared__ uint16_t s_list[STHREADS];
	__shared__ uint32_t s_stats[3];   // [0] k-mer starts resolved (= probes of the reference), [1] list elements, [2] hash probes
	__shared__ uint16_t s_q[QCAP];    // k-mer starts the lanes' one anchor + walk left over: (item << 8) | position in the pass
	__shared__ uint32_t s_qn;

	const DevDB &db = A.db;
	const int tid = threadIdx.x;
	const int k = (int) db.kmersize;
	const int64_t n_active = (int64_t) A.counters[A.in_count];
	// first tier: one group of items per workgroup; second tier: a fixed grid that loops over the (usually few) groups
	int64_t first = (int64_t) blockIdx.x * SG;
	if(first >= n_active) return;
	do {
	const int ng = (int) min((int64_t) SG, n_active - first);

	if(tid < SG) {
		int L = 0, nN = 0; int64_t so = 0, no = 0, it = 0;
		int dg = (int) DIAG_NONE;
		if(tid < ng) {
			const int64_t raw = A.in_items[first + tid];
			it = raw & ITEM_MASK;
			if(A.cat_bases > 0 && (raw >> 33) != 0) dg = (int) (raw >> 33);          // (the overflow lists carry bare items)
			const int64_t r = it >> 1;
			L = A.len[r]; so = A.seq_off[r]; no = A.N_off[r]; nN = (int) (A.N_off[r + 1] - no);
		}
		s_len[tid] = L; s_nN[tid] = nN; s_soff[tid] = so; s_noff[tid] = no; s_item[tid] = it; s_diag[tid] = dg;
		// the longest read of the group, among the 16 lanes that hold the items (one DPP row: no LDS round, no barrier of its own)
		int mx = L;
#pragma unroll
		for(int d = SG / 2; d; d >>= 1) mx = max(mx, __shfl_xor(mx, d, SG));
		if(tid == 0) { s_gmax = mx - k + 1; s_anylong = mx > (SW - 1) * 32; s_qn = 0; }
	}
	if(tid < 3) s_stats[tid] = 0;
	__syncthreads();
	const bool staged_once = !s_anylong;
	const int gmax = s_gmax;
	// the words are staged in STRAND orientation (word w = strand bases 32w .. 32w+31: for the reverse strand the reverse
	// complement of the forward words), so the passes below never reverse-complement a k-mer or a walk window
	if(staged_once) {
		for(int idx = tid; idx < SG * SW; idx += STHREADS) {
			const int g = idx / SW, w = idx - g * SW;
			const int L = s_len[g];
			uint64_t v = 0;
			if(w < ((L + 31) >> 5)) v = (s_item[g] & 1) ? strand_win(A.seq + s_soff[g], L, 1, w << 5) : A.seq[s_soff[g] + w];
			w_lds[idx] = v;
		}
	}

	// Scores are computed per template, not per position: the score machine of save_kmers
	// (savekmers.c:2511-2706) gives template t   k*M  at its first hit and  bridge(p - 1 - last_t)
	// at every later hit p (a hit = the value set of the k-mer at p holds t); runs of equal sets and
	// set changes are only how the reference walks that sum. So every (item, run of equal sets)
	// ORs its position range into the bitmask of each template of the set, and every
	// (item, template) then folds its own bitmask -- no serial walk over the positions.
	{
		for(int idx = tid; idx < TSLOTS * SG; idx += STHREADS) {
			t_id[idx] = T_EMPTY; t_score[idx] = INT_MIN;          // (t_last / t_first are only read behind a score)
#pragma unroll
			for(int w = 0; w < MW; ++w) t_mask[w * TSLOTS * SG + idx] = 0;
		}
		for(int idx = tid; idx < VSLOTS * SG; idx += STHREADS) {
			v_id[idx] = MISS;
#pragma unroll
			for(int w = 0; w < MW; ++w) v_mask[w * VSLOTS * SG + idx] = 0;
		}
		if(tid < SG) { t_cnt[tid] = 0; s_over[tid] = 0; s_hits[tid] = 0; s_best[tid] = 0; s_nb[tid] = 0; }
		__syncthreads();

		for(int c0 = 0; c0 < gmax; c0 += CHUNK) {
			const bool more_passes = c0 + CHUNK < gmax;      // (every group starts from cleared tables)
			// stage the forward words of this pass (only workgroups holding a read too long to be staged once)
			if(!staged_once) {
				for(int idx = tid; idx < SG * SW; idx += STHREADS) {
					const int g = idx / SW, w = idx - g * SW;
					uint64_t v = 0;
					if(g < ng) {
						const int L = s_len[g];
						const int wb = c0 >> 5;
						if(wb + w < ((L + 31) >> 5)) v = (s_item[g] & 1) ? strand_win(A.seq + s_soff[g], L, 1, (wb + w) << 5) : A.seq[s_soff[g] + wb + w];
					}
					w_lds[idx] = v;
				}
				__syncthreads();
			}
			// phase 1: value set of every k-mer start of the pass. A read that matches a template keeps matching it, so
			// each of the 16 lanes of an item anchors its 9-position segment with ONE hash probe and then walks along
			// the concatenated template store: while the next read base equals the next template base, the next
			// k-mer is the template's next k-mer and its value-list offset is a sequential 4-byte read of vs_id.
			// Only anchors and the positions after a disagreement cost a random gather into the probe table.
			// Every run of equal value lists inside a walk ORs its position range into the mask of that LIST in the
			// item's v-table -- no list is read here, and a list that recurs along the read is expanded once.
			uint32_t nprobe = 0, nres = 0;
			// one run of equal value lists, positions [rs, re) of the pass (re - rs <= SEG): into the item's v-table
			auto add_run = [&](int g, uint32_t vi, int rs, int re) {
				if(STATS) atomicAdd(&s_stats[1], value_at(db, vi, 0) + 1u);
				// claim / find the list's slot in the item's v-table
				int slot = -1;
				const uint32_t h = (vi * 0x9E3779B1u) >> 28;
#pragma unroll 1
				for(int x = 0; x < VSLOTS; ++x) {
					const int sidx = (int) ((h + x) & (VSLOTS - 1)) * SG + g;
					const uint32_t old = atomicCAS(&v_id[sidx], MISS, vi);
					if(old == MISS || old == vi) { slot = sidx; break; }
				}
				if(slot >= 0) {
					// a run inside one lane's segment is at most SEG (9) positions long: two mask words at most
					const int w = rs >> 5;
					const uint64_t m = ((1ull << (re - rs)) - 1ull) << (rs & 31);
					atomicOr(&v_mask[w * VSLOTS * SG + slot], (uint32_t) m);
					if(m >> 32) atomicOr(&v_mask[(w + 1) * VSLOTS * SG + slot], (uint32_t) (m >> 32));
				} else {
					// more distinct lists in this pass than the v-table holds: expand this run directly
					expand_list<TSLOTS>(db, vi, rs, re, g, t_id, t_cnt, t_mask, s_over);
				}
			};
			// one k-mer start on its own (position jj of the pass, item g): probe, and a run of one position when it hits
			auto resolve = [&](int g, int jj) {
				const int L = s_len[g], strand = (int) (s_item[g] & 1), nN = s_nN[g];
				const int p = c0 + jj;
				const int q = strand ? (L - k - p) : p;
				if(nN && window_has_N(A.N + s_noff[g], nN, q, k)) return;
				const uint64_t *wsrc = &w_lds[g * SW];
				const int w = (p >> 5) - (staged_once ? 0 : (c0 >> 5));
				const uint64_t km = kmer_from(wsrc[w], wsrc[w + 1], p, k);
				uint32_t gp;
#ifdef KMAHIP_DIAG
				if(A.ablate & 2) gp = MISS; else
#endif
				gp = probe(db, (uint32_t) km);
				++nprobe; ++nres;
				if(gp == MISS) return;
				const uint32_t vi = db.vs_id[gp];
				if(MODE) atomicAdd(&s_hits[g], 1);
#ifdef KMAHIP_DIAG
				if(A.ablate & 1) return;
#endif
				if(!s_over[g]) add_run(g, vi, jj, jj + 1);
			};
			int own_lo = 0, own_n = 0;          // k-mer starts this lane could not queue
			{
				constexpr int LPI = STHREADS / SG;                 // lanes per item
				constexpr int SEG = (CHUNK + LPI - 1) / LPI;         // positions per lane
				const int g = tid & (SG - 1), sl = tid / SG;
				const int j0 = sl * SEG, j1 = min(CHUNK, j0 + SEG);
				// ONE anchor probe and walk per lane, straight-line. What a lane has left afterwards -- the k-mer starts behind a
				// miss or behind a walk that a mismatch cut short -- goes into a workgroup-wide queue and is resolved below by all
				// threads side by side, one probe each: a lane that resolved them itself, two per round trip, kept its workgroup
				// waiting for five dependent gathers (profiles/r3_scan_phase_counters.md: phase 1 was 28 % of the instructions
				// and 47 % of the time).
				if(g < ng && c0 + j0 < s_len[g] - k + 1) {
					const int L = s_len[g], npos = L - k + 1, strand = (int) (s_item[g] & 1), nN = s_nN[g];
					const int32_t *Nl = A.N + s_noff[g];
					const uint64_t *wsrc = &w_lds[g * SW];
					const int wb = staged_once ? 0 : (c0 >> 5);
					int jj = j0, hc = 0;
					const int p = c0 + jj;
					const int q = strand ? (L - k - p) : p;       // forward coordinate of the window (the N list is forward)
					if(!(nN && window_has_N(Nl, nN, q, k))) {
						const int w = (p >> 5) - wb;
						const uint64_t km = kmer_from(wsrc[w], wsrc[w + 1], p, k);
						const uint64_t qw = win2(wsrc, p - (wb << 5));          // the k-mer and the 32 - k bases behind it
						// Where is this k-mer in `cat`? First on the diagonal of the prefilter's hit (a read that maps lies on ONE template
						// diagonal: no gather into the probe table, and the 16 lanes of an item read neighbouring bytes of cat / vs_id), then
						// by its hash probe. A position is right when the k bases there are the k-mer and a k-mer of the index starts there.
						uint32_t gp = MISS;
						int attempt = 1;
						const int dg = s_diag[g];
						if(dg != (int) DIAG_NONE) {
							const int64_t at = (int64_t) dg - DIAG_BIAS + p;
							if(at >= 0 && at < A.cat_bases) { gp = (uint32_t) at; attempt = 0; }
						}
						for(;;) {
							if(attempt) {
#ifdef KMAHIP_DIAG
								if(A.ablate & 2) gp = MISS; else
#endif
								gp = probe(db, (uint32_t) km);
								++nprobe;
								if(gp == MISS) break;
							}
							// everything the walk needs depends only on gp: issue it all at once (one latency, not one per step)
							constexpr int WALK = SEG - 1;
							static_assert(WALK + 16 <= 32, "one 32-base window holds the k-mer (k <= 16) and the walk behind it");
							uint32_t vv[WALK + 1];
							const uint32_t *vp = db.vs_id + gp;          // one address, immediate offsets: the loads merge
#pragma unroll
							for(int i = 0; i <= WALK; ++i) vv[i] = vp[i];
							const uint64_t tw = win2(db.cat, (int64_t) gp);
							const uint64_t x = qw ^ tw;
							const int same = x ? (__clzll((long long) x) >> 1) : 32;
							if(!attempt && !(same >= k && vv[0] != KMAHIP_EMPTY_VI)) {
								attempt = 1; continue;          // not on that diagonal: probe
							}
							// walk: how many more k-mer starts of this segment continue the same template diagonal
							int run = 0;
							int room = min(j1 - jj - 1, npos - (p + 1));
							if(nN && room > 0) {
								// the walk may not run into an N: strand position of the first N at or after p + k
								int lo = 0, hi = nN;
								if(!strand) { while(lo < hi) { const int mid = (lo + hi) >> 1; if(Nl[mid] < p + k) lo = mid + 1; else hi = mid; }
									if(lo < nN) room = min(room, Nl[lo] - (p + k)); }
								else { const int fq = L - 1 - (p + k);      // forward position of strand base p + k; bases go down from here
									while(lo < hi) { const int mid = (lo + hi) >> 1; if(Nl[mid] <= fq) lo = mid + 1; else hi = mid; }
									if(lo > 0) room = min(room, fq - Nl[lo - 1]); }
							}
							if(room > 0) {
								run = min(room, same - k);
								// the template ends where vs_id holds no k-mer: nothing after it continues the diagonal
#pragma unroll
								for(int i = 1; i <= WALK; ++i) if(i <= run && vv[i] == KMAHIP_EMPTY_VI) run = i - 1;
							}
							// positions jj .. jj + run carry the lists vv[0 .. run]; bit i of bm: a run of equal lists starts at i
							uint32_t bm = 1u;
#pragma unroll
							for(int i = 1; i <= WALK; ++i) if(i <= run && vv[i] != vv[i - 1]) bm |= 1u << i;
							nres += run; hc += run + 1;
#ifdef KMAHIP_DIAG
							if(A.ablate & 1) bm = 0;
#endif
							if(s_over[g]) bm = 0;                // the item goes to the overflow kernel anyway
							while(bm) {
								const int i0 = __ffs((int) bm) - 1;
								bm &= bm - 1;
								const int i1 = bm ? __ffs((int) bm) - 1 : run + 1;
								uint32_t vi = vv[0];
#pragma unroll
								for(int i = 1; i <= WALK; ++i) if(i0 == i) vi = vv[i];
								add_run(g, vi, jj + i0, jj + i1);            // positions [rs, re) of the pass
							}
							jj += run;
							break;
						}
						++nres;
					}
					++jj;
					// what is left of the segment: into the queue; when the queue is full (every read of the group riddled with
					// mismatches) the lane keeps it and resolves it itself behind the queue round
					const int left = min(j1, npos - c0) - jj;
					if(left > 0) {
						const uint32_t at = atomicAdd(&s_qn, (uint32_t) left);
						if(at + (uint32_t) left <= (uint32_t) QCAP) for(int i = 0; i < left; ++i) s_q[at + i] = (uint16_t) ((g << 8) | (jj + i));
						else {
							for(int i = 0; i < left; ++i) if(at + i < (uint32_t) QCAP) s_q[at + i] = 0xFFFFu;
							own_lo = jj; own_n = left;
						}
					}
					if(MODE && hc) atomicAdd(&s_hits[g], hc);
				}
			}
			__syncthreads();
			{
				// the queue: every entry one k-mer start (item, position in the pass), one probe each
				const uint32_t qn = min(s_qn, (uint32_t) QCAP);
				const int q_iters = (int) ((qn + STHREADS - 1) / STHREADS);
				// (a lane that kept its k-mer starts takes them one by one behind the queue's rounds)
				for(int it = 0; it < q_iters + own_n; ++it) {
					uint32_t v = 0xFFFFu;
					if(it < q_iters) { const uint32_t e = (uint32_t) it * STHREADS + tid; if(e < qn) v = s_q[e]; }
					else v = (uint32_t) (((tid & (SG - 1)) << 8) | (own_lo + it - q_iters));
					if(v != 0xFFFFu) resolve((int) (v >> 8), (int) (v & 255u));
				}
			}
			if(STATS && nres) { atomicAdd(&s_stats[0], nres); atomicAdd(&s_stats[2], nprobe); }
			__syncthreads();
			if(tid == 0) s_qn = 0;          // (for the next pass / the next group: barriers lie in between)
			// phase 2a: one thread per (item, distinct value list): read the list ONCE (all gathers of the workgroup in
			// flight together) and OR the list's position mask into the hit mask of each listed template
			static_assert((VSLOTS * SG) % STHREADS == 0 && (TSLOTS * SG) % STHREADS == 0, "table slots per thread");
			for(int part = 0; part < VSLOTS * SG; part += STHREADS) {
			const int n_lists = compact_threads(v_id[part + tid] != MISS, tid, s_wcnt, s_list);
			if(tid < n_lists) {
				const int idx = part + s_list[tid];
				const int g = idx & (SG - 1);
				const uint32_t vi = v_id[idx];
				uint32_t mw[MW];
#pragma unroll
				for(int w = 0; w < MW; ++w) mw[w] = v_mask[w * VSLOTS * SG + idx];
				if(more_passes) {          // (the tables of the last pass of a first-tier workgroup are not looked at again)
					v_id[idx] = MISS;
#pragma unroll
					for(int w = 0; w < MW; ++w) v_mask[w * VSLOTS * SG + idx] = 0;
				}
				if(g < ng && !s_over[g]) {
				// list head: count + 7 ids in flight together (the value arrays carry 8 pad elements)
				uint32_t cnt, el[7];
				if(db.values_u16) {
					const uint16_t *vp = db.values16 + vi;
					cnt = vp[0];
#pragma unroll
					for(int i = 0; i < 7; ++i) el[i] = vp[1 + i];
				} else {
					const uint32_t *vp = db.values32 + vi;
					cnt = vp[0];
#pragma unroll
					for(int i = 0; i < 7; ++i) el[i] = vp[1 + i];
				}
				auto add_template = [&](const uint32_t t) -> bool {
					const int slot = template_slot<TSLOTS>(t, g, t_id, t_cnt);
					if(slot < 0) { s_over[g] = 1; return false; }
#pragma unroll
					for(int w = 0; w < MW; ++w) if(mw[w]) atomicOr(&t_mask[w * TSLOTS * SG + slot], mw[w]);
					return true;
				};
				bool ok = true;
#pragma unroll 1
				for(uint32_t i = 0; ok && i < cnt; ++i) {
					uint32_t t = el[0];
#pragma unroll
					for(int e = 1; e < 7; ++e) if(i == (uint32_t) e) t = el[e];
					if(i >= 7) t = value_at(db, vi, (int) i + 1);
					ok = add_template(t);
				}
				}
			}
			}
			__syncthreads();
			// phase 2b: one thread per (item, template): fold the hit mask into the score
			for(int part = 0; part < TSLOTS * SG; part += STHREADS) {
			const int n_tmpl = compact_threads((tid & (SG - 1)) < ng && t_id[part + tid] != T_EMPTY, tid, s_wcnt, s_list);
#ifdef KMAHIP_DIAG
			if(!(A.ablate & 8))
#endif
			if(tid < n_tmpl) {
				const int idx = part + s_list[tid];
				// (the first pass starts every slot afresh: nothing to read)
				int score = INT_MIN, last = 0, first = 0;
				if(c0) { score = t_score[idx]; last = t_last[idx]; first = t_first[idx]; }
				// the 136-bit mask as three 64-bit words; a run of ones starts where a one has a zero below it and ends
				// (exclusively) where a zero has a one below it -- MW * 32 > CHUNK, so every run ends inside the words
				static_assert(MW == 5 && CHUNK < MW * 32, "mask layout");
				uint32_t mw[MW];
#pragma unroll
				for(int w = 0; w < MW; ++w) mw[w] = t_mask[w * TSLOTS * SG + idx];
				const uint64_t M0 = (uint64_t) mw[0] | ((uint64_t) mw[1] << 32), M1 = (uint64_t) mw[2] | ((uint64_t) mw[3] << 32), M2 = mw[4];
				const uint64_t X0 = M0 << 1, X1 = (M1 << 1) | (M0 >> 63), X2 = (M2 << 1) | (M1 >> 63);
				uint64_t S0 = M0 & ~X0, S1 = M1 & ~X1, S2 = M2 & ~X2;
				uint64_t E0 = ~M0 & X0, E1 = ~M1 & X1, E2 = ~M2 & X2;
				if(M0 | M1 | M2) {
					// every hit but the first of a run is worth M; the first one k*M (first hit ever) or the bridge over the
					// zeros before it -- so only the zero runs BETWEEN runs are walked (none for the usual single run)
					const int ones = __popcll(M0) + __popcll(M1) + __popcll(M2);
					const int nruns = __popcll(S0) + __popcll(S1) + __popcll(S2);
					const int fb = M0 ? __ffsll((long long) M0) - 1 : M1 ? 63 + __ffsll((long long) M1) : 127 + __ffsll((long long) M2);
					const int lb = M2 ? 191 - __clzll((long long) M2) : M1 ? 127 - __clzll((long long) M1) : 63 - __clzll((long long) M0);
					if(score == INT_MIN) { score = k * A.M; first = c0 + fb; }
					else score += bridge(c0 + fb - 1 - last, k, A.M, A.MM, A.U, A.W1);
					score += (ones - nruns) * A.M;
					last = c0 + lb;
					// drop the start of the first run; then the i-th end pairs with the start of run i + 1
					if(S0) S0 &= S0 - 1; else if(S1) S1 &= S1 - 1; else S2 &= S2 - 1;
					for(int r = 1; r < nruns; ++r) {
						int b0, b1;
						if(S0) { b0 = __ffsll((long long) S0) - 1; S0 &= S0 - 1; }
						else if(S1) { b0 = 63 + __ffsll((long long) S1); S1 &= S1 - 1; }
						else { b0 = 127 + __ffsll((long long) S2); S2 &= S2 - 1; }
						if(E0) { b1 = __ffsll((long long) E0) - 1; E0 &= E0 - 1; }
						else if(E1) { b1 = 63 + __ffsll((long long) E1); E1 &= E1 - 1; }
						else { b1 = 127 + __ffsll((long long) E2); E2 &= E2 - 1; }
						score += bridge(b0 - b1, k, A.M, A.MM, A.U, A.W1);      // b1 = first zero after a run, b0 = next run's start
					}
				}
				t_score[idx] = score; t_first[idx] = first;
				if(more_passes) {
					t_last[idx] = last;
#pragma unroll
					for(int w = 0; w < MW; ++w) t_mask[w * TSLOTS * SG + idx] = 0;
				}
			}
			}
			__syncthreads();
		}

		// ---- finish the items of this group: getBestMatch (savekmers.c:273-294), all threads: the best score of an item by an LDS
		// atomic over its occupied slots, the tied templates counted the same way, one thread per item takes the result slots, and
		// every tied template finds its place among the others -- first-seen order = ascending (first hit position, template id) --
		// by counting the smaller keys (one lane per item walking the 16 slots once per tie kept the other three waves waiting)
		static_assert((TSLOTS * SG) % STHREADS == 0, "table slots per thread");
		for(int part = 0; part < TSLOTS * SG; part += STHREADS) {
			const int idx = part + tid, g = idx & (SG - 1);
			if(g < ng && !s_over[g] && t_id[idx] != T_EMPTY) {
				if(MODE) atomicAdd(&s_nb[g], 1);
				else atomicMax(&s_best[g], max(0, t_score[idx]));
			}
		}
		__syncthreads();
		if(!MODE) {
			for(int part = 0; part < TSLOTS * SG; part += STHREADS) {
				const int idx = part + tid, g = idx & (SG - 1);
				if(g < ng && !s_over[g] && t_id[idx] != T_EMPTY) { const int sc = max(0, t_score[idx]); if(sc > 0 && sc == s_best[g]) atomicAdd(&s_nb[g], 1); }
			}
			__syncthreads();
		}
		if(tid < SG) {
			// (the SG lanes that hold the items: lanes 0 .. SG - 1 of the first wavefront)
			const int g = tid;
			const bool mine = g < ng;
			const int64_t item = s_item[g];
			int best = 0, nb = 0, want = 0;
			bool over = false;
#ifdef KMAHIP_DIAG
			if(A.ablate & 16) { nb = 0; } else
#endif
			if(mine && s_over[g]) { over = true; nb = -1; }
			else if(mine) {
				// MODE 1, get_kmers_for_pair (savekmers.c:427-688): all candidates, first-seen order, clamped scores, the hit count
				best = MODE ? s_hits[g] : s_best[g];
				nb = (MODE || best > 0) ? s_nb[g] : 0;
				if(nb && (MODE || nb > INL)) want = nb;
			}
			// pool room for the lists that do not fit the inline slots, and places in the overflow list: ONE atomic per workgroup and
			// counter (an atomic per item on one address -- every item in the all-candidates mode of paired reads -- took more than the scan)
			int incl = want, oincl = over ? 1 : 0;
#pragma unroll
			for(int d = 1; d < SG; d <<= 1) {
				const int a = __shfl_up(incl, d, SG), b = __shfl_up(oincl, d, SG);
				if((tid & (SG - 1)) >= d) { incl += a; oincl += b; }
			}
			const int tot = __shfl(incl, SG - 1, SG), otot = __shfl(oincl, SG - 1, SG);
			unsigned long long pbase = 0, obase = 0;
			// all-candidates mode, first tier: every item has a list, and even one atomic per workgroup on the pool's one counter is
			// 6 ns a workgroup -- as long as the whole scan took. The lists take fixed places instead (the item's place in the active
			// list x the most a first-tier table holds) whenever the pool has the room; the second tier and the best-templates mode
			// (a list longer than the inline slots is rare there) keep the counter.
			const bool fixed = MODE && TSLOTS == TS1 && n_active * (int64_t) (TSLOTS - 2) <= A.pool_cap - A.pool_tail0;
			if(fixed) {
				pbase = (unsigned long long) first * (TSLOTS - 2); incl = tid * (TSLOTS - 2) + want;
				if(first == 0 && tid == 0) atomicAdd(&A.counters[C_POOL], (unsigned long long) (n_active * (int64_t) (TSLOTS - 2)));      // (the later tiers allocate behind the fixed places)
			}
			else if(tid == 0 && tot) pbase = atomicAdd(&A.counters[C_POOL], (unsigned long long) tot);
			if(tid == 0 && otot) obase = atomicAdd(&A.counters[A.out_count], (unsigned long long) otot);
			if(!fixed) pbase = __shfl(pbase, 0, SG);
			obase = __shfl(obase, 0, SG);
			int64_t off = 0;
			if(over) { A.out_over[obase + (unsigned long long) (oincl - 1)] = item; s_off[g] = -1; }
			else if(mine) {
				if(nb) off = want ? A.pool_tail0 + (int64_t) pbase + (incl - want) : item * INL;
				if(nb && off + nb > A.pool_cap) { atomicMax(&A.counters[C_STATUS], 1ull); s_off[g] = -1; }
				else s_off[g] = nb ? off : -1;
			}
			if(mine) {
				A.item_score[item] = best;
				A.item_n[item] = nb;
				A.item_off[item] = off;
			}
		}
		__syncthreads();
		for(int part = 0; part < TSLOTS * SG; part += STHREADS) {
			const int idx = part + tid, g = idx & (SG - 1);
			if(g >= ng || s_off[g] < 0 || t_id[idx] == T_EMPTY) continue;
			const int sc = max(0, t_score[idx]), best = s_best[g];
			if(!MODE && sc != best) continue;
			int rank = 0;
			if(s_nb[g] > 1) {
				const long long key = ((long long) t_first[idx] << 32) | t_id[idx];
#pragma unroll 1
				for(int x = 0; x < TSLOTS; ++x) {
					const int j = x * SG + g;
					const uint32_t id = t_id[j];
					if(id == T_EMPTY || (!MODE && max(0, t_score[j]) != best)) continue;
					rank += ((((long long) t_first[j] << 32) | id) < key);
				}
			}
			A.pool[s_off[g] + rank] = (int32_t) t_id[idx];
			if(MODE) A.pool_sc[s_off[g] + rank] = sc;
		}
		__syncthreads();
	}
	if(STATS) {
		if(tid == 0) {
			atomicAdd(&A.counters[C_PROBES], (unsigned long long) s_stats[0]);
			atomicAdd(&A.counters[C_VALS], (unsigned long long) s_stats[1]);
			atomicAdd(&A.counters[C_HASH], (unsigned long long) s_stats[2]);
		}
		__syncthreads();
	}
	if(TSLOTS == TS1) break;          // (straight-line code in the first tier)
	first += (int64_t) gridDim.x * SG;
	} while(first < n_active);
}

// Overflow path: items whose candidate set does not fit the LDS tables. The reference's own sequential formulation
// (savekmers.c:2511-2706) on DB_size-wide score / last-hit / list arrays in HBM (its per-thread layout, :134-150), one
// WAVEFRONT per item: the 64 lanes resolve 512 k-mer starts at a time side by side (the probes are what the old
// one-lane version spent its time waiting for), then walk them in lockstep -- the walk itself is uniform, and the
// templates of a value list are handled one per lane.
constexpr int DCH = 512;

__global__ __launch_bounds__(64) void scan_dense_kernel(const ScanArgs A) {
	__shared__ uint32_t s_vi[DCH];
	const DevDB &db = A.db;
	const int k = (int) db.kmersize, lane = threadIdx.x;
	const int64_t n_over = (int64_t) A.counters[A.in_count];
	const int64_t slot = blockIdx.x;
	if(slot >= A.dense_slots) return;
	const int64_t D = db.DB_size;
	volatile int32_t *score = A.dense + slot * 3 * D;
	volatile int32_t *ext = score + D;
	volatile int32_t *list = ext + D;   // list[0..nlist): templates in first-seen order; ext[t] == -1 marks "absent"
	const unsigned long long below = (1ull << lane) - 1ull;
	for(int64_t oi = slot; oi < n_over; oi += A.dense_slots) {
		const int64_t item = A.in_items[oi];
		const int64_t r = item >> 1;
		const int strand = (int) (item & 1);
		const int L = A.len[r], npos = L - k + 1;
		const uint64_t *rs = A.seq + A.seq_off[r];
		const int32_t *Nl = A.N + A.N_off[r];
		const int nN = (int) (A.N_off[r + 1] - A.N_off[r]);
		uint32_t last = NONE;
		int gaps = 0, HIT = 0, acc = 0, nlist = 0, hits = 0;
		// every template of list vl, one per lane: f(t, first-seen index or -1)
		for(int base = 0; base < npos; base += DCH) {
			const int n = min(DCH, npos - base);
			for(int x = lane; x < n; x += 64) {
				const int p = base + x, q = strand ? (L - k - p) : p;
				uint32_t vi = MISS;
				if(nN == 0 || !window_has_N(Nl, nN, q, k)) {
					const int w = q >> 5;
					uint64_t km = kmer_from(rs[w], rs[w + 1], q, k);
					if(strand) km = revcomp_kmer(km, k);
					vi = probe(db, (uint32_t) km);
					if(vi != MISS) vi = db.vs_id[vi];
				}
				s_vi[x] = vi;
			}
			__syncthreads();
			for(int x = 0; x < n; ++x) {
				const int p = base + x;
				const uint32_t vi = s_vi[x];
				if(vi == MISS) { ++gaps; continue; }
				if(vi == last) {
					acc += bridge(gaps, k, A.M, A.MM, A.U, A.W1);
				} else {
					if(last != NONE) {
						const int c = (int) value_at(db, last, 0);
						for(int i = 1 + lane; i <= c; i += 64) { const uint32_t t = value_at(db, last, i); score[t] += acc; ext[t] = HIT; }
						__threadfence();
					}
					HIT = p - 1;
					const int cnt = (int) value_at(db, vi, 0);
					for(int i0 = 1; i0 <= cnt; i0 += 64) {
						const int i = i0 + lane;
						bool fresh = false;
						uint32_t t = 0;
						if(i <= cnt) {
							t = value_at(db, vi, i);
							if(ext[t] != -1) score[t] += bridge(HIT - ext[t], k, A.M, A.MM, A.U, A.W1);
							else { score[t] = k * A.M; ext[t] = 0; fresh = true; }
						}
						const unsigned long long fm = __ballot(fresh);
						if(fresh) list[nlist + __popcll(fm & below)] = (int32_t) t;
						nlist += __popcll(fm);
					}
					__threadfence();
					last = vi;
					acc = 0;
				}
				HIT = p; gaps = 0; ++hits;
			}
			__syncthreads();
		}
		int best = 0, nb = 0;
		int64_t off = 0;
		if(hits) {
			const int c = (int) value_at(db, last, 0);
			for(int i = 1 + lane; i <= c; i += 64) score[value_at(db, last, i)] += acc;
			__threadfence();
		}
		if(hits && A.mode) {
			best = hits; nb = nlist;
			if(lane == 0) off = A.pool_tail0 + (int64_t) atomicAdd(&A.counters[C_POOL], (unsigned long long) nb);
			off = __shfl(off, 0);
			if(off + nb <= A.pool_cap) {
				for(int e = lane; e < nlist; e += 64) { A.pool[off + e] = list[e]; A.pool_sc[off + e] = max(0, (int) score[list[e]]); }
			} else if(lane == 0) atomicMax(&A.counters[C_STATUS], 1ull);
		} else if(hits) {
			for(int e = lane; e < nlist; e += 64) best = max(best, max(0, (int) score[list[e]]));
			for(int d = 32; d; d >>= 1) best = max(best, __shfl_xor(best, d));
			if(best > 0) {
				for(int e = lane; e < nlist; e += 64) nb += max(0, (int) score[list[e]]) == best;
				for(int d = 32; d; d >>= 1) nb += __shfl_xor(nb, d);
				if(lane == 0) off = A.pool_tail0 + (int64_t) atomicAdd(&A.counters[C_POOL], (unsigned long long) nb);
				off = __shfl(off, 0);
				if(off + nb <= A.pool_cap) {
					int w = 0;
					for(int e0 = 0; e0 < nlist; e0 += 64) {
						const int e = e0 + lane;
						const bool is = e < nlist && max(0, (int) score[list[e]]) == best;
						const unsigned long long m = __ballot(is);
						if(is) A.pool[off + w + __popcll(m & below)] = list[e];
						w += __popcll(m);
					}
				} else if(lane == 0) atomicMax(&A.counters[C_STATUS], 1ull);
			} else nb = 0;
		}
		for(int e = lane; e < nlist; e += 64) { const int t = list[e]; score[t] = 0; ext[t] = -1; }
		__threadfence();
		if(lane == 0) { A.item_score[item] = best; A.item_n[item] = nb; A.item_off[item] = off; }
	}
}

// ---- combine: strand decision + CSR ---------------------------------------
constexpr int CB = 256;

__global__ __launch_bounds__(CB) void combine_count_kernel(const ScanArgs A, int32_t *rc_flag, int32_t *flag, int64_t *T_off, int64_t *blk_sums) {
	__shared__ int64_t red[CB];
	const int64_t r = (int64_t) blockIdx.x * CB + threadIdx.x;
	int64_t nT = 0;
	if(r < A.n_reads) {
		const int k = (int) A.db.kmersize;
		const int bs = A.item_score[2 * r], br = A.item_score[2 * r + 1];
		int rf = 0, fl = 0;
		// savekmers.c:3037-3062
		if((bs > 0 || br > 0) && (k <= bs || k <= br)) {
			if(bs > br) { nT = A.item_n[2 * r]; rf = bs; }
			else if(bs < br) { nT = A.item_n[2 * r + 1]; rf = br; fl = 16; }
			else { nT = A.item_n[2 * r] + A.item_n[2 * r + 1]; rf = -bs; }
		}
		rc_flag[r] = rf;
		flag[r] = fl;
		T_off[r + 1] = nT; // per-read count for now
	}
	red[threadIdx.x] = nT;
	__syncthreads();
	for(int s = CB / 2; s > 0; s >>= 1) {
		if(threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
		__syncthreads();
	}
	if(threadIdx.x == 0) blk_sums[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(1024) void scan_blocks_kernel(int64_t *blk_sums, int64_t nblk) {
	// single workgroup exclusive scan over the per-block totals
	__shared__ int64_t part[1024];
	const int t = threadIdx.x;
	const int64_t per = (nblk + 1023) / 1024;
	const int64_t b0 = t * per, b1 = min(nblk, b0 + per);
	int64_t s = 0;
	for(int64_t i = b0; i < b1; ++i) s += blk_sums[i];
	part[t] = s;
	__syncthreads();
	for(int d = 1; d < 1024; d <<= 1) {
		int64_t v = (t >= d) ? part[t - d] : 0;
		__syncthreads();
		part[t] += v;
		__syncthreads();
	}
	int64_t run = part[t] - s;
	for(int64_t i = b0; i < b1; ++i) { const int64_t v = blk_sums[i]; blk_sums[i] = run; run += v; }
}

__global__ __launch_bounds__(CB) void combine_write_kernel(const ScanArgs A, const int32_t *rc_flag, const int32_t *flag,
                                                            int64_t *T_off, const int64_t *blk_sums, int32_t *T, int64_t T_cap) {
	__shared__ int64_t sc[CB];
	const int t = threadIdx.x;
	const int64_t r = (int64_t) blockIdx.x * CB + t;
	const int64_t nT = (r < A.n_reads) ? T_off[r + 1] : 0;
	sc[t] = nT;
	__syncthreads();
	for(int d = 1; d < CB; d <<= 1) {
		int64_t v = (t >= d) ? sc[t - d] : 0;
		__syncthreads();
		sc[t] += v;
		__syncthreads();
	}
	if(r >= A.n_reads) return;
	const int64_t end = blk_sums[blockIdx.x] + sc[t];
	const int64_t beg = end - nT;
	T_off[r + 1] = end;
	if(r == 0) T_off[0] = 0;
	if(nT == 0) return;
	if(end > T_cap) { atomicMax(&A.counters[C_STATUS], 2ull); return; }
	const int rf = rc_flag[r], fl = flag[r];
	int64_t w = beg;
	if(rf > 0 && fl == 0) {
		const int64_t o = A.item_off[2 * r];
		for(int i = 0; i < A.item_n[2 * r]; ++i) T[w++] = A.pool[o + i];
	} else if(rf > 0) {
		const int64_t o = A.item_off[2 * r + 1];
		for(int i = 0; i < A.item_n[2 * r + 1]; ++i) T[w++] = A.pool[o + i];
	} else {
		int64_t o = A.item_off[2 * r];
		for(int i = 0; i < A.item_n[2 * r]; ++i) T[w++] = A.pool[o + i];
		o = A.item_off[2 * r + 1];
		for(int i = 0; i < A.item_n[2 * r + 1]; ++i) T[w++] = -A.pool[o + i];
	}
}

// dense scratch planes per slot: [score = 0][last-hit = -1 (absent)][list]
__global__ void dense_init_kernel(int32_t *p, int64_t n, int64_t D) {
	int64_t i = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t stride = (int64_t) gridDim.x * blockDim.x;
	for(; i < n; i += stride) p[i] = ((i / D) % 3 == 1) ? -1 : 0;
}

// ---- paired end, `-apm p` ------------------------------------------------------------------
// save_kmers_penaltyPair (savekmers.c:3572-3777) over the four strand items of a pair
// (mate1 fwd/rc = items 4p, 4p+1; mate2 = 4p+2, 4p+3): getFirstPen (:1383), getSecondBestPen (:1415),
// getF_Best (:1648). One lane per pair; lists live in the item pool, results in the pair pool.
struct PList { const int32_t *t, *s; int n; };

__device__ __forceinline__ int plist_find(const PList &l, int t) {
	for(int i = 0; i < l.n; ++i) if(l.t[i] == t) return l.s[i];
	return 0;
}

struct PairArgs {
	ScanArgs S;
	int64_t n_pairs;
	int PE;
	int32_t *ppool;          // pair pool: record template lists
	int64_t ppool_cap;
	// per record (2 per pair, stream order)
	int32_t *r_mate, *r_rc, *r_score, *r_flag, *r_n;
	int64_t *r_off;
};

__global__ __launch_bounds__(256) void pair_penalty_kernel(const PairArgs P) {
	const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(p >= P.n_pairs) return;
	const ScanArgs &A = P.S;
	const int k = (int) A.db.kmersize;
	PList L[4];
	int hc[4];
	for(int x = 0; x < 4; ++x) {
		const int64_t it = 4 * p + x;
		L[x].n = max(0, A.item_n[it]); L[x].t = A.pool + A.item_off[it]; L[x].s = A.pool_sc + A.item_off[it];
		hc[x] = A.item_score[it];
	}
	const PList &F1 = L[0], &R1 = L[1], &F2 = L[2], &R2 = L[3];
	const int hc1 = max(hc[0], hc[1]), hc2 = max(hc[2], hc[3]);
	const int len1 = A.len[2 * p], len2 = A.len[2 * p + 1];
	// working lists in the pair pool
	const int n1 = F1.n + R1.n, n2 = F2.n + R2.n;
	const int64_t need = (int64_t) max(n1, n2) + n2;
	int64_t base = 0;
	if(need) {
		base = (int64_t) atomicAdd(&A.counters[C_PPOOL], (unsigned long long) need);
		if(base + need > P.ppool_cap) { atomicMax(&A.counters[C_STATUS], 1ull); return; }
	}
	int32_t *regT = P.ppool + base, *bT = regT + max(n1, n2);
	auto regS = [&](int i) { return i < F1.n ? F1.s[i] : R1.s[i - F1.n]; };
	int nreg = 0, best1 = 0;
	if(hc1) {
		for(int i = 0; i < F1.n; ++i) { best1 = max(best1, F1.s[i]); regT[nreg++] = F1.t[i]; }
		for(int i = 0; i < R1.n; ++i) { best1 = max(best1, R1.s[i]); regT[nreg++] = -R1.t[i]; }
	}
	int paired = 0, best2 = 0, nb2 = 0;
	if(hc2) {
		if(0 < best1) {
			for(int i = 0; i < F2.n; ++i) { best2 = max(best2, F2.s[i]); bT[nb2++] = F2.t[i]; }
			for(int i = 0; i < R2.n; ++i) { best2 = max(best2, R2.s[i]); bT[nb2++] = -R2.t[i]; }
			int hits = 0;
			if(best2) {
				int comp = max(0, best1 + best2 - P.PE);
				for(int i = 0; i < nreg; ++i) {
					const int rt = regT[i];
					int sc = rt > 0 ? plist_find(R2, rt) : plist_find(F2, -rt);
					if(0 < sc) {
						sc += regS(i);
						if(comp < sc) { comp = sc; hits = 1; regT[0] = rt; }
						else if(comp == sc) { regT[hits++] = rt; }
					}
				}
			}
			if(hits) { paired = 1; nreg = hits; }
			else {
				int c = 0;
				for(int i = 0; i < nreg; ++i) if(best1 == regS(i)) regT[c++] = regT[i];
				nreg = c; c = 0;
				for(int i = 0; i < nb2; ++i) {
					const int t = bT[i];
					if(0 < t) { if(best2 == plist_find(F2, t)) bT[c++] = t; }
					else { if(best2 <= plist_find(R2, -t)) bT[c++] = t; }
				}
				nb2 = c;
			}
		} else {
			nreg = 0;
			for(int i = 0; i < F2.n; ++i) { const int sc = F2.s[i]; if(best2 < sc) { best2 = sc; nreg = 0; regT[nreg++] = F2.t[i]; } else if(best2 == sc) regT[nreg++] = F2.t[i]; }
			for(int i = 0; i < R2.n; ++i) { const int sc = R2.s[i]; if(best2 < sc) { best2 = sc; nreg = 0; regT[nreg++] = -R2.t[i]; } else if(best2 == sc) regT[nreg++] = -R2.t[i]; }
		}
	}
	int o1 = len1 >= k, o2 = len2 >= k;      // get_kmers_for_pair leaves a scanned mate reverse-complemented
	int flag = 65, flag_r = 129;
	int m[2] = {-1, -1}, rcv[2] = {0, 0}, sc[2] = {0, 0}, fl[2] = {0, 0}, nn[2] = {0, 0};
	int64_t of[2] = {0, 0};
	// CompDNA.seqlen is unsigned, so the coverage tests below wrap like the reference's
	if(0 < best1 && 0 < best2) {
		if(paired) {
			flag |= 2; flag_r |= 2;
			const int comp = min(hc1 + hc2, best1 + best2);
			if(k <= comp || (unsigned) (len1 + len2 - comp - (k << 1)) < (unsigned) (comp * k)) {
				if(0 < regT[0]) {
					flag |= 32; flag_r |= 16; o1 ^= 1;
					m[0] = 0; rcv[0] = o1; sc[0] = best1; fl[0] = flag; nn[0] = 0;
					m[1] = 1; rcv[1] = o2; sc[1] = best2; fl[1] = flag_r; nn[1] = nreg; of[1] = base;
				} else {
					flag |= 16; flag_r |= 32; o2 ^= 1;
					for(int i = 0; i < nreg; ++i) regT[i] = -regT[i];
					m[0] = 1; rcv[0] = o2; sc[0] = best2; fl[0] = flag_r; nn[0] = 0;
					m[1] = 0; rcv[1] = o1; sc[1] = best1; fl[1] = flag; nn[1] = nreg; of[1] = base;
				}
			}
		} else {
			const int h1 = min(hc1, best1), h2 = min(hc2, best2);
			const bool ok1 = k <= h1 || (unsigned) (len1 - h1 - k) < (unsigned) (h1 * k);
			const bool ok2 = k <= h2 || (unsigned) (len2 - h2 - k) < (unsigned) (h2 * k);
			int s1 = best1, s2 = best2;
			if(ok1) {
				if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
				else { flag |= 16; flag_r |= 32; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			}
			if(ok2) {
				if(0 < bT[0]) { o2 ^= 1; if(bT[nb2 - 1] < 0) s2 = -s2; }
				else { flag |= 32; flag_r |= 16; for(int i = 0; i < nb2; ++i) bT[i] = -bT[i]; }
			}
			if(ok1) { m[0] = 0; rcv[0] = o1; sc[0] = s1; fl[0] = flag; nn[0] = nreg; of[0] = base; }
			if(ok2) { m[1] = 1; rcv[1] = o2; sc[1] = s2; fl[1] = flag_r; nn[1] = nb2; of[1] = base + max(n1, n2); }
		}
	} else if(0 < best1) {
		const int h1 = min(hc1, best1);
		int s1 = best1;
		if(k <= h1 || (unsigned) (len1 - h1 - k) < (unsigned) (h1 * k)) {
			flag |= 8; flag |= 32;
			if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
			else { flag |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			m[0] = 0; rcv[0] = o1; sc[0] = s1; fl[0] = flag; nn[0] = nreg; of[0] = base;
		}
	} else if(0 < best2) {
		const int h2 = min(hc2, best2);
		int s2 = best2;
		if(k <= h2 || (unsigned) (len2 - h2 - k) < (unsigned) (h2 * k)) {
			flag_r |= 8; flag_r |= 32;
			if(0 < regT[0]) { o2 ^= 1; if(regT[nreg - 1] < 0) s2 = -s2; }
			else { flag_r |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			m[1] = 1; rcv[1] = o2; sc[1] = s2; fl[1] = flag_r; nn[1] = nreg; of[1] = base;
		}
	}
	for(int x = 0; x < 2; ++x) {
		const int64_t r = 2 * p + x;
		P.r_mate[r] = m[x]; P.r_rc[r] = rcv[x]; P.r_score[r] = sc[x]; P.r_flag[r] = fl[x];
		P.r_n[r] = (m[x] >= 0) ? nn[x] : 0; P.r_off[r] = of[x];
	}
}

// `-apm u` (and `-ipe` without -apm: the reference's default, kma.c:206): save_kmers_unionPair, savekmers.c:3367-3570, with getF_Best /
// getR_Best (:1648-1762). Mate 1 keeps its best-scoring templates of either strand; mate 2 its own -- and where a template of mate 1's
// set is also in mate 2's on the OTHER strand the two are a couple on those templates (moved to the front of the list, in the order
// met); else each mate is a record of its own. Same inputs and outputs as pair_penalty_kernel.
__global__ __launch_bounds__(256) void pair_union_kernel(const PairArgs P) {
	const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(p >= P.n_pairs) return;
	const ScanArgs &A = P.S;
	const int k = (int) A.db.kmersize;
	PList L[4];
	int hc[4];
	for(int x = 0; x < 4; ++x) {
		const int64_t it = 4 * p + x;
		L[x].n = max(0, A.item_n[it]); L[x].t = A.pool + A.item_off[it]; L[x].s = A.pool_sc + A.item_off[it];
		hc[x] = A.item_score[it];
	}
	const PList &F1 = L[0], &R1 = L[1], &F2 = L[2], &R2 = L[3];
	const int hc1 = max(hc[0], hc[1]), hc2 = max(hc[2], hc[3]);
	const int len1 = A.len[2 * p], len2 = A.len[2 * p + 1];
	const int n1 = F1.n + R1.n, n2 = F2.n + R2.n;
	const int64_t need = (int64_t) max(n1, n2) + n2;
	int64_t base = 0;
	if(need) {
		base = (int64_t) atomicAdd(&A.counters[C_PPOOL], (unsigned long long) need);
		if(base + need > P.ppool_cap) { atomicMax(&A.counters[C_STATUS], 1ull); return; }
	}
	int32_t *regT = P.ppool + base, *bT = regT + max(n1, n2);
	// getF_Best: the best score over both strands' candidates and the templates that reach it, forward ones first
	auto best_of = [&](const PList &F, const PList &R, int32_t *dst, int &cnt) {
		int best = 0;
		cnt = 0;
		for(int i = 0; i < F.n; ++i) { const int sc = F.s[i]; if(best < sc) { best = sc; cnt = 0; dst[cnt++] = F.t[i]; } else if(best == sc) dst[cnt++] = F.t[i]; }
		for(int i = 0; i < R.n; ++i) { const int sc = R.s[i]; if(best < sc) { best = sc; cnt = 0; dst[cnt++] = -R.t[i]; } else if(best == sc) dst[cnt++] = -R.t[i]; }
		return best;
	};
	// (CompDNA.seqlen is unsigned: the coverage test wraps like the reference's)
	auto covered = [&](int best, int len) { return !(k < best && (unsigned) (best * k) < (unsigned) len - (unsigned) best); };
	int nreg = 0, nb2 = 0, best1 = 0, best2 = 0, paired = 0;
	if(hc1) {
		best1 = best_of(F1, R1, regT, nreg);
		if(!covered(best1, len1)) best1 = 0;
	}
	if(hc2) {
		if(best1) {
			// getR_Best: mate 2's own best set, then the templates of mate 1's set that are in it on the other strand
			best2 = best_of(F2, R2, bT, nb2);
			int hits = 0;
			if(0 < best2) {
				for(int i = 0; i < nreg; ++i) {
					const int rt = regT[i];
					const int sc = rt > 0 ? plist_find(R2, rt) : plist_find(F2, -rt);
					if(sc == best2) { const int x = regT[hits]; regT[hits] = rt; regT[i] = x; ++hits; }
				}
			}
			if(hits) { paired = 1; nreg = hits; }
		} else best2 = best_of(F2, R2, regT, nreg);
		if(!covered(best2, len2)) { best2 = 0; paired = 0; }
	}
	int o1 = len1 >= k, o2 = len2 >= k;      // get_kmers_for_pair leaves a scanned mate reverse-complemented
	int flag = 65, flag_r = 129;
	int m[2] = {-1, -1}, rcv[2] = {0, 0}, sc[2] = {0, 0}, fl[2] = {0, 0}, nn[2] = {0, 0};
	int64_t of[2] = {0, 0};
	if(0 < best1 && 0 < best2) {
		if(paired) {
			flag |= 2; flag_r |= 2;
			if(0 < regT[0]) {
				flag |= 32; flag_r |= 16; o1 ^= 1;
				m[0] = 0; rcv[0] = o1; sc[0] = best1; fl[0] = flag; nn[0] = 0;
				m[1] = 1; rcv[1] = o2; sc[1] = best2; fl[1] = flag_r; nn[1] = nreg; of[1] = base;
			} else {
				flag |= 16; flag_r |= 32; o2 ^= 1;
				for(int i = 0; i < nreg; ++i) regT[i] = -regT[i];
				m[0] = 1; rcv[0] = o2; sc[0] = best2; fl[0] = flag_r; nn[0] = 0;
				m[1] = 0; rcv[1] = o1; sc[1] = best1; fl[1] = flag; nn[1] = nreg; of[1] = base;
			}
		} else {
			int s1 = best1, s2 = best2;
			if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
			else { flag |= 16; flag_r |= 32; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
			if(0 < bT[0]) { o2 ^= 1; if(bT[nb2 - 1] < 0) s2 = -s2; }
			else { flag |= 32; flag_r |= 16; for(int i = 0; i < nb2; ++i) bT[i] = -bT[i]; }
			m[0] = 0; rcv[0] = o1; sc[0] = s1; fl[0] = flag; nn[0] = nreg; of[0] = base;
			m[1] = 1; rcv[1] = o2; sc[1] = s2; fl[1] = flag_r; nn[1] = nb2; of[1] = base + max(n1, n2);
		}
	} else if(best1) {
		int s1 = best1;
		flag |= 8; flag |= 32;
		if(0 < regT[0]) { o1 ^= 1; if(regT[nreg - 1] < 0) s1 = -s1; }
		else { flag |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
		m[0] = 0; rcv[0] = o1; sc[0] = s1; fl[0] = flag; nn[0] = nreg; of[0] = base;
	} else if(best2) {
		int s2 = best2;
		flag_r |= 8; flag_r |= 32;
		if(0 < regT[0]) { o2 ^= 1; if(regT[nreg - 1] < 0) s2 = -s2; }
		else { flag_r |= 16; for(int i = 0; i < nreg; ++i) regT[i] = -regT[i]; }
		m[1] = 1; rcv[1] = o2; sc[1] = s2; fl[1] = flag_r; nn[1] = nreg; of[1] = base;
	}
	for(int x = 0; x < 2; ++x) {
		const int64_t r = 2 * p + x;
		P.r_mate[r] = m[x]; P.r_rc[r] = rcv[x]; P.r_score[r] = sc[x]; P.r_flag[r] = fl[x];
		P.r_n[r] = (m[x] >= 0) ? nn[x] : 0; P.r_off[r] = of[x];
	}
}

// `-apm f` / `-pm f`: save_kmers_forcePair, savekmers.c:3779-3864, with getFirstForce (:1254) / getSecondBestForce (:1275). Mate 1's
// candidates of both strands with their scores; those that mate 2 also hits on the OTHER strand, at the best sum of the two scores, are
// the couple's templates -- or there is no record at all: forced pairing files nothing singly. Both records of a couple carry the sum
// (negated when the list ends on a reverse template). Same inputs and outputs as pair_penalty_kernel.
__global__ __launch_bounds__(256) void pair_force_kernel(const PairArgs P) {
	const int64_t p = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(p >= P.n_pairs) return;
	const ScanArgs &A = P.S;
	const int k = (int) A.db.kmersize;
	PList L[4];
	int hc[4];
	for(int x = 0; x < 4; ++x) {
		const int64_t it = 4 * p + x;
		L[x].n = max(0, A.item_n[it]); L[x].t = A.pool + A.item_off[it]; L[x].s = A.pool_sc + A.item_off[it];
		hc[x] = A.item_score[it];
	}
	const PList &F1 = L[0], &R1 = L[1], &F2 = L[2], &R2 = L[3];
	const int hc1 = max(hc[0], hc[1]), hc2 = max(hc[2], hc[3]);
	const int len1 = A.len[2 * p], len2 = A.len[2 * p + 1];
	int m[2] = {-1, -1}, rcv[2] = {0, 0}, sc[2] = {0, 0}, fl[2] = {0, 0}, nn[2] = {0, 0};
	int64_t of[2] = {0, 0};
	const int n1 = F1.n + R1.n;
	if(hc1 && hc2 && n1) {          // (without a hit of mate 1 the reference does not even scan mate 2: no record either way)
		const int64_t base = (int64_t) atomicAdd(&A.counters[C_PPOOL], (unsigned long long) n1);
		if(base + n1 > P.ppool_cap) { atomicMax(&A.counters[C_STATUS], 1ull); return; }
		int32_t *regT = P.ppool + base;
		// getSecondBestForce over getFirstForce's list (mate 1's forward candidates, then its reverse ones as negative ids)
		int best = 0, hits = 0;
		for(int i = 0; i < n1; ++i) {
			const bool fw = i < F1.n;
			const int t = fw ? F1.t[i] : R1.t[i - F1.n], s1 = fw ? F1.s[i] : R1.s[i - F1.n];
			const int s2 = fw ? plist_find(R2, t) : plist_find(F2, t);
			if(!s2) continue;
			const int sum = s1 + s2;
			if(best < sum) { best = sum; hits = 0; regT[hits++] = fw ? t : -t; }
			else if(best == sum) regT[hits++] = fw ? t : -t;
		}
		// (CompDNA.seqlen is unsigned: the coverage test wraps like the reference's)
		if(best && (k <= best || (unsigned) len1 + (unsigned) len2 - (unsigned) best < (unsigned) (best * k))) {
			int o1 = len1 >= k, o2 = len2 >= k;      // get_kmers_for_pair leaves a scanned mate reverse-complemented
			int flag = 67, flag_r = 131;
			const int s = regT[hits - 1] < 0 ? -best : best;
			if(0 < regT[0]) {
				flag |= 32; flag_r |= 16; o1 ^= 1;
				m[0] = 0; rcv[0] = o1; sc[0] = s; fl[0] = flag; nn[0] = 0;
				m[1] = 1; rcv[1] = o2; sc[1] = s; fl[1] = flag_r; nn[1] = hits; of[1] = base;
			} else {
				flag |= 16; flag_r |= 32; o2 ^= 1;
				for(int i = 0; i < hits; ++i) regT[i] = -regT[i];
				m[0] = 1; rcv[0] = o2; sc[0] = s; fl[0] = flag_r; nn[0] = 0;
				m[1] = 0; rcv[1] = o1; sc[1] = s; fl[1] = flag; nn[1] = hits; of[1] = base;
			}
		}
	}
	for(int x = 0; x < 2; ++x) {
		const int64_t r = 2 * p + x;
		P.r_mate[r] = m[x]; P.r_rc[r] = rcv[x]; P.r_score[r] = sc[x]; P.r_flag[r] = fl[x];
		P.r_n[r] = (m[x] >= 0) ? nn[x] : 0; P.r_off[r] = of[x];
	}
}

// generic CSR compaction of per-record lists: counts -> offsets (3 kernels)
__global__ __launch_bounds__(CB) void rec_count_kernel(const int32_t *cnt, int64_t n, int64_t *blk_sums) {
	__shared__ int64_t red[CB];
	const int64_t r = (int64_t) blockIdx.x * CB + threadIdx.x;
	red[threadIdx.x] = (r < n) ? cnt[r] : 0;
	__syncthreads();
	for(int s = CB / 2; s > 0; s >>= 1) {
		if(threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
		__syncthreads();
	}
	if(threadIdx.x == 0) blk_sums[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(CB) void rec_write_kernel(const int32_t *cnt, const int64_t *src_off, const int32_t *src, int64_t n,
                                                        const int64_t *blk_sums, int64_t *out_off, int32_t *out, int64_t out_cap,
                                                        unsigned long long *counters) {
	__shared__ int64_t sc[CB];
	const int t = threadIdx.x;
	const int64_t r = (int64_t) blockIdx.x * CB + t;
	const int64_t c = (r < n) ? cnt[r] : 0;
	sc[t] = c;
	__syncthreads();
	for(int d = 1; d < CB; d <<= 1) {
		int64_t v = (t >= d) ? sc[t - d] : 0;
		__syncthreads();
		sc[t] += v;
		__syncthreads();
	}
	if(r >= n) return;
	const int64_t end = blk_sums[blockIdx.x] + sc[t], beg = end - c;
	out_off[r + 1] = end;
	if(r == 0) out_off[0] = 0;
	if(c == 0) return;
	if(end > out_cap) { atomicMax(&counters[C_STATUS], 2ull); return; }
	const int64_t so = src_off[r];
	for(int64_t i = 0; i < c; ++i) out[beg + i] = src[so + i];
}


// ---- anchors of the default mode's chain finder, side by side (save_kmers_chain, savekmers.c:5208-5452; the sequential form is
// build_ankers in chain.hip) ---------------------------------------------------------------------------------------------------
// An anchor is a maximal run of k-mer starts whose value list is the same as that of the hit before them, at a distance of 0 or
// exactly k missed starts. One workgroup = 16 strand items that passed the prefilter x 16 lanes:
//   1. every lane resolves the value list of its 9 k-mer starts per pass with the anchor + walk of scan_se_kernel and leaves the
//      list offsets in LDS, indexed by FORWARD read position (the reverse strand's anchors are counted in forward coordinates);
//   2. the 16 lanes of an item (one DPP row now) turn them into anchors: the hit before a lane's segment by a "last non-empty"
//      scan across the row, open / continue per hit, the anchor numbers by a prefix sum of the opens, weights and last hits by LDS
//      atomics (a run may span several lanes);
//   3. the anchors go to a device-wide pool, one allocation per workgroup.
// Reads with N's (where the reference's reverse strand restarts k bases off, savekmers.c:5447-5449), with more than CA_NPMAX k-mer
// starts or more than CA_AMAX anchors on a strand are left to the other routes of chain.hip (slow[read] = 1).
// (anchors per strand the tables hold: 32 -- at 64 the tables took 8 kB more of LDS and the kernel ran four workgroups per CU instead of
// five, 3.4 ms per 2 M reads instead of 3.0; 14 reads in 2 M carry more and take the long-read route, at 20 it is three in a hundred)
#ifndef KMAHIP_CA_AMAX
#define KMAHIP_CA_AMAX 32
#endif
constexpr int CA_SEG = 9, CA_PASS = GROUP * CA_SEG, CA_NPMAX = 2 * CA_PASS, CA_AMAX = KMAHIP_CA_AMAX, CA_WORDS = 12, CA_VSTRIDE = CA_NPMAX + 1;

struct AnchorArgs {
	ScanArgs S;
	KmaAnk *pool;
	int64_t pool_cap;
	int64_t *a_off;
	int32_t *a_n;
	uint8_t *slow;
	unsigned long long *cnt;      // [0] anchors allocated
	int vstride;                  // words per item of the list table (dynamic LDS): the batch's most k-mer starts + 1, odd, CA_VSTRIDE at most
};

__global__ __launch_bounds__(THREADS, 4) void chain_anchor_kernel(const AnchorArgs A) {
	// value list per forward k-mer start (MISS: none), row stride odd. Sized by the batch's longest read, not by the longest the kernel
	// takes: with 150-base reads 9 kB instead of 18, eight workgroups per CU instead of five (the kernel waits for its lookups)
	extern __shared__ uint32_t s_val[];
	const int VSTRIDE = A.vstride;
	__shared__ uint64_t s_w[GROUP * CA_WORDS];              // the read in strand orientation
	// (an item's anchors side by side, the items CA_AMAX + 1 apart: as [anchor][item] the lanes of an item -- consecutive anchors -- fell on
	// four banks, SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE)
	__shared__ uint32_t a_val[(CA_AMAX + 1) * GROUP], a_start[(CA_AMAX + 1) * GROUP], a_last[(CA_AMAX + 1) * GROUP];
	__shared__ int32_t a_w[(CA_AMAX + 1) * GROUP];
	__shared__ int32_t s_npos[GROUP], s_cnt[GROUP], s_off[GROUP];
	__shared__ int64_t s_item[GROUP];
	__shared__ unsigned long long s_base;
	const ScanArgs &S = A.S;
	const DevDB &db = S.db;
	const int tid = threadIdx.x, k = (int) db.kmersize;
	const int64_t n_active = (int64_t) S.counters[C_NACT];
	const int64_t first = (int64_t) blockIdx.x * GROUP;
	if(first >= n_active) return;
	const int ng = (int) min((int64_t) GROUP, n_active - first);
	for(int i = tid; i < (CA_AMAX + 1) * GROUP; i += THREADS) { a_w[i] = 0; a_last[i] = 0; }
	{
		// phase 1 layout: item g = tid & 15, lane sl = tid >> 4 (as in scan_se_kernel: a wave's gathers belong to 16 reads)
		const int g = tid & (GROUP - 1), sl = tid / GROUP;
		int64_t item = 0, r = 0, so = 0;
		int L = 0, npos = 0, strand = 0;
		bool live = g < ng;
		if(live) {
			item = S.in_items[first + g] & ITEM_MASK;
			r = item >> 1; strand = (int) (item & 1);
			L = S.len[r]; npos = L - k + 1; so = S.seq_off[r];
			if(S.N_off[r + 1] != S.N_off[r] || npos > CA_NPMAX || npos >= VSTRIDE || npos <= 0) { if(sl == 0) A.slow[r] = 1; live = false; }
		}
		if(sl == 0) { s_npos[g] = live ? npos : 0; s_item[g] = item; }
		if(sl < CA_WORDS) {
			uint64_t v = 0;
			if(live && sl < ((L + 31) >> 5)) v = strand ? strand_win(S.seq + so, L, 1, sl << 5) : S.seq[so + sl];
			s_w[g * CA_WORDS + sl] = v;
		}
		__syncthreads();
		const uint64_t *wsrc = &s_w[g * CA_WORDS];
		uint32_t *vrow = &s_val[g * VSTRIDE];
		auto put = [&](int p, uint32_t vi) { vrow[strand ? npos - 1 - p : p] = vi; };
		for(int c0 = 0; c0 < CA_NPMAX; c0 += CA_PASS) {
			const int j0 = c0 + sl * CA_SEG, j1 = j0 + CA_SEG;
			if(!live || j0 >= npos) continue;
			int jj = j0;
			bool pairs = false;
			while(jj < j1 && jj < npos) {
				int p = jj;
				const uint64_t km = kmer_from(wsrc[p >> 5], wsrc[(p >> 5) + 1], p, k);
				uint32_t gp;
				if(pairs && jj + 1 < j1 && p + 1 < npos) {
					const uint64_t km2 = kmer_from(wsrc[(p + 1) >> 5], wsrc[((p + 1) >> 5) + 1], p + 1, k);
					uint32_t gp2;
					probe2(db, (uint32_t) km, (uint32_t) km2, gp, gp2);
					if(gp == MISS) {
						put(p, MISS);
						if(gp2 == MISS) { put(p + 1, MISS); jj += 2; continue; }
						gp = gp2; ++jj; ++p;
					}
				} else gp = probe(db, (uint32_t) km);
				if(gp == MISS) { put(p, MISS); ++jj; pairs = true; continue; }
				constexpr int WALK = CA_SEG - 1;
				uint32_t vv[WALK + 1];
				const uint32_t *vp = db.vs_id + gp;
#pragma unroll
				for(int i = 0; i <= WALK; ++i) vv[i] = vp[i];
				const uint64_t tw = win2(db.cat, (int64_t) gp + k);
				const uint64_t qw = win2(wsrc, p + k);
				int run = 0;
				const int room = min(j1 - jj - 1, npos - (p + 1));
				if(room > 0) {
					const uint64_t x = qw ^ tw;
					const int same = x ? (__clzll((long long) x) >> 1) : 32;
					run = min(room, same);
#pragma unroll
					for(int i = 1; i <= WALK; ++i) if(i <= run && vv[i] == KMAHIP_EMPTY_VI) run = i - 1;
				}
#pragma unroll
				for(int i = 0; i <= WALK; ++i) if(i <= run) put(p + i, vv[i]);
				jj += run + 1;
				pairs = true;
			}
		}
	}
	__syncthreads();
	// phase 2 layout: item gi = tid >> 4, lane ln = tid & 15 (the lanes of an item are one DPP row)
	const int gi = tid >> 4, ln = tid & 15;
	const int npos = s_npos[gi];              // 0: nothing to do for this item
	const uint32_t *vrow = &s_val[gi * VSTRIDE];
	int carry_h = -1, carry_cnt = 0;
	uint32_t carry_v = MISS;
	for(int c0 = 0; c0 < CA_NPMAX; c0 += CA_PASS) {
		const int j0 = c0 + ln * CA_SEG;
		uint32_t v[CA_SEG];
		int lh = -1;
		uint32_t lv = MISS;
#pragma unroll
		for(int i = 0; i < CA_SEG; ++i) {
			v[i] = (j0 + i < npos) ? vrow[j0 + i] : MISS;
			if(v[i] != MISS) { lh = j0 + i; lv = v[i]; }
		}
		// the last hit at or before each lane's segment (inclusive scan of "last non-empty"), then the one before it
		int h = lh;
		uint32_t hv = lv;
#pragma unroll
		for(int d = 1; d < 16; d <<= 1) {
			const int oh = __shfl_up(h, d, 16);
			const uint32_t ov = __shfl_up(hv, d, 16);
			if(ln >= d && h < 0) { h = oh; hv = ov; }
		}
		int ph = __shfl_up(h, 1, 16);
		uint32_t pv = __shfl_up(hv, 1, 16);
		if(ln == 0 || ph < 0) { ph = carry_h; pv = carry_v; }
		// open / continue per hit (savekmers.c:5262-5300: the same list as the hit before, 0 or exactly k starts missed in between)
		int opens = 0, th = ph;
		uint32_t tv = pv, code = 0;
#pragma unroll
		for(int i = 0; i < CA_SEG; ++i) {
			if(v[i] == MISS) continue;
			const int j = j0 + i;
			uint32_t c = 1;
			if(th >= 0 && v[i] == tv) { const int gaps = j - th - 1; if(gaps == 0) c = 2; else if(gaps == k) c = 3; }
			if(c == 1) ++opens;
			code |= c << (2 * i);
			th = j; tv = v[i];
		}
		int inc = opens;
#pragma unroll
		for(int d = 1; d < 16; d <<= 1) { const int o = __shfl_up(inc, d, 16); if(ln >= d) inc += o; }
		// (a lane's hits of one anchor summed in registers: two LDS atomics per anchor and lane, not per hit)
		int idx = carry_cnt + inc - opens - 1, accw = 0, accl = -1;
#pragma unroll
		for(int i = 0; i < CA_SEG; ++i) {
			const uint32_t c = (code >> (2 * i)) & 3u;
			if(!c) continue;
			const int j = j0 + i;
			if(c == 1) {
				if(accl >= 0 && idx < CA_AMAX) { atomicAdd(&a_w[gi * (CA_AMAX + 1) + idx], accw); atomicMax(&a_last[gi * (CA_AMAX + 1) + idx], (uint32_t) accl); }
				++idx;
				if(idx < CA_AMAX) { a_start[gi * (CA_AMAX + 1) + idx] = (uint32_t) j; a_val[gi * (CA_AMAX + 1) + idx] = v[i]; }
				accw = k * S.M;
			} else accw += c == 2 ? S.M : k * S.M + S.MM;
			accl = j;
		}
		if(accl >= 0 && idx < CA_AMAX) { atomicAdd(&a_w[gi * (CA_AMAX + 1) + idx], accw); atomicMax(&a_last[gi * (CA_AMAX + 1) + idx], (uint32_t) accl); }
		carry_cnt += __shfl(inc, 15, 16);
		const int eh = __shfl(h, 15, 16);
		const uint32_t ev = __shfl(hv, 15, 16);
		if(eh >= 0) { carry_h = eh; carry_v = ev; }
	}
	const int n = carry_cnt;
	const bool over = n > CA_AMAX;
	if(ln == 0) s_cnt[gi] = over ? 0 : n;
	__syncthreads();
	if(tid == 0) {
		int tot = 0;
		for(int x = 0; x < GROUP; ++x) { s_off[x] = tot; tot += s_cnt[x]; }
		s_base = tot ? atomicAdd(&A.cnt[0], (unsigned long long) tot) : 0ull;
	}
	__syncthreads();
	if(npos <= 0 || gi >= ng) return;
	const int64_t item = s_item[gi];
	if(over) { if(ln == 0) A.slow[item >> 1] = 1; return; }
	const int64_t off = (int64_t) s_base + s_off[gi];
	if(ln == 0) { A.a_off[item] = off; A.a_n[item] = n; }
	if(off + n > A.pool_cap) return;          // (the host sees cnt[0] > pool_cap and repeats with a larger pool)
	const int L = npos + k - 1;
	(void) L;
	for(int i = ln; i < n; i += 16) {
		KmaAnk a;
		const uint32_t last = a_last[gi * (CA_AMAX + 1) + i];
		const uint32_t vi = a_val[gi * (CA_AMAX + 1) + i];
		// the head of the anchor's value list rides in the two fields the chaining keeps in registers (score_len: the length,
		// and with 16-bit lists the first element in its upper half; len_len: the next two, or the first 32-bit one): the lane
		// that chains the read (chain_fast_kernel) then needs no gather per anchor and listed template for lists of up to three
		if(db.values_u16) {
			const uint16_t *vp = db.values16 + vi;          // (the value arrays carry 8 pad elements)
			const uint32_t nl = vp[0], e1 = vp[1], e2 = vp[2], e3 = vp[3];
			a.score_len = (int) (nl | (e1 << 16)); a.len_len = (int) (e2 | (e3 << 16));
		} else {
			a.score_len = (int) db.values32[vi]; a.len_len = (int) db.values32[vi + 1];
		}
		a.score = 0; a.weight = a_w[gi * (CA_AMAX + 1) + i];
		a.start = a_start[gi * (CA_AMAX + 1) + i];
		// an anchor closed by the next one ends behind its last hit's k-mer + 1 (j - gaps + k at the opening hit j); the last one
		// at seqlen - gaps with the k missed starts of the read's end counted in: its last hit (savekmers.c:5316-5330)
		a.end = i < n - 1 ? last + 1u + (uint32_t) k : last;
		a.values = vi;
		a.descend = i < n - 1 ? i + 1 : -1;
		A.pool[off + i] = a;
	}
}
} // namespace


// the diagonals of the prefilter's hits are filed (ScanArgs::cat_bases) when `cat` is small enough for the list entry's field
// (KMAHIP_SCAN_DIAG=0: never)
static int64_t diag_cat_bases(const kmahip_db *db) {
	if(const char *e = getenv("KMAHIP_SCAN_DIAG")) if(!atoi(e)) return 0;
	const int64_t total = db->h_cat_off.empty() ? 0 : db->h_cat_off.back();
	return total > 0 && total < (1ll << 30) ? total : 0;
}

static int ws_reserve(kmahip_ws *ws, int64_t n_reads) {
	kmahip_db *db = ws->db;
	if(n_reads > ws->cap_reads) {
		(void) hipFree(ws->item_score); (void) hipFree(ws->item_n); (void) hipFree(ws->item_off);
		(void) hipFree(ws->pool); (void) hipFree(ws->overflow_items); (void) hipFree(ws->blk_sums); (void) hipFree(ws->active_items);
		ws->item_score = ws->item_n = nullptr; ws->item_off = nullptr; ws->pool = nullptr;
		ws->overflow_items = nullptr; ws->blk_sums = nullptr; ws->active_items = nullptr;
		const int64_t cap = n_reads + n_reads / 8 + 1024;
		HIP_TRY(hipMalloc((void **) &ws->item_score, cap * 2 * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->item_n, cap * 2 * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->item_off, cap * 2 * sizeof(int64_t)));
		if(ws->pool_scale < 1) ws->pool_scale = 1;
		ws->pool_cap = cap * 16 * ws->pool_scale;
		HIP_TRY(hipMalloc((void **) &ws->pool, ws->pool_cap * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->overflow_items, cap * 4 * sizeof(int64_t)));       // first-tier list + second-tier list
		HIP_TRY(hipMalloc((void **) &ws->active_items, cap * 2 * sizeof(int64_t)));
		ws->blk_cap = (cap + CB - 1) / CB + 1;
		HIP_TRY(hipMalloc((void **) &ws->blk_sums, ws->blk_cap * sizeof(int64_t)));
		ws->cap_reads = cap;
	}
	if(!ws->counters) { HIP_TRY(hipMalloc((void **) &ws->counters, N_COUNTERS * sizeof(unsigned long long))); HIP_TRY(hipMemset(ws->counters, 0, N_COUNTERS * sizeof(unsigned long long))); }
	if(!ws->dense) {
		// overflow scratch: up to 4096 concurrent items, bounded to 1 GiB
		int64_t slots = 4096;
		const int64_t per = (int64_t) db->info.DB_size * 3 * sizeof(int32_t);
		while(slots > 64 && slots * per > (1ll << 30)) slots >>= 1;
		ws->dense_slots = slots;
		HIP_TRY(hipMalloc((void **) &ws->dense, slots * per));
		hipLaunchKernelGGL(dense_init_kernel, dim3(1024), dim3(256), 0, 0, ws->dense, slots * 3 * (int64_t) db->info.DB_size, (int64_t) db->info.DB_size);
		HIP_TRY(hipDeviceSynchronize());
	}
	return KMAHIP_OK;
}

int kmahip_launch_scan_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                          const kmahip_params *p, kmahip_cands *out, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	if(n < 0 || !out || !p) { kmahip_set_error("bad arguments"); return KMAHIP_EINVAL; }
	int rc = ws_reserve(ws, n > 0 ? n : 1);
	if(rc) return rc;
	ScanArgs A;
	A.db = db->dev;
	A.n_reads = n; A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1; A.exhaustive = p->exhaustive;
	A.item_score = ws->item_score; A.item_n = ws->item_n; A.item_off = ws->item_off;
	A.pool = ws->pool; A.pool_cap = ws->pool_cap; A.counters = ws->counters; A.overflow_items = ws->overflow_items;
	A.dense = ws->dense; A.dense_slots = ws->dense_slots; A.active_items = ws->active_items;
	A.mode = 0; A.pool_sc = nullptr; A.pool_tail0 = 2 * n * INL; A.cat_bases = diag_cat_bases(db);
	A.ablate = 0;
#ifdef KMAHIP_DIAG
	if(const char *e = getenv("KMAHIP_ABLATE_SCAN")) A.ablate = atoi(e);
#endif
	// word 1 (status) is sticky until kmahip_ws_status reads it
	HIP_TRY(hipMemsetAsync(ws->counters, 0, sizeof(unsigned long long), stream));
	HIP_TRY(hipMemsetAsync(ws->counters + 2, 0, (N_COUNTERS - 2) * sizeof(unsigned long long), stream));
	if(n == 0) {
		HIP_TRY(hipMemsetAsync(out->T_off, 0, sizeof(int64_t), stream));
		return KMAHIP_OK;
	}
	const int64_t items = 2 * n;
	const unsigned pgrid = (unsigned) ((items + PF_ITEMS - 1) / PF_ITEMS);
	const unsigned grid = (unsigned) ((items + SG - 1) / SG);      // scan: upper bound; workgroups past the active count exit
	hipEvent_t ev0 = nullptr, ev1 = nullptr, evp = nullptr, evq = nullptr;
	if(ws->timing_on) {
		HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventCreate(&evp)); HIP_TRY(hipEventCreate(&evq));
		HIP_TRY(hipEventRecord(evp, stream));
	}
	if(ws->stats_on) hipLaunchKernelGGL((scan_prefilter_kernel<true>), dim3(pgrid), dim3(THREADS), 0, stream, A);
	else hipLaunchKernelGGL((scan_prefilter_kernel<false>), dim3(pgrid), dim3(THREADS), 0, stream, A);
	if(ws->timing_on) { HIP_TRY(hipEventRecord(evq, stream)); HIP_TRY(hipEventRecord(ev0, stream)); }
	int64_t *over1 = ws->overflow_items, *over2 = ws->overflow_items + 2 * ws->cap_reads;
	A.in_items = ws->active_items; A.in_count = C_NACT; A.out_over = over1; A.out_count = C_NOVER;
	if(ws->stats_on) hipLaunchKernelGGL((scan_se_kernel<true, 0, TS1>), dim3(grid), dim3(STHREADS), 0, stream, A);
	else hipLaunchKernelGGL((scan_se_kernel<false, 0, TS1>), dim3(grid), dim3(STHREADS), 0, stream, A);
	if(ws->timing_on) {
		HIP_TRY(hipEventRecord(ev1, stream));
		if(!ws->events) ws->events = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
		ws->events->push_back({ev0, ev1});
		if(!ws->events3) ws->events3 = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
		ws->events3->push_back({evp, evq});
	}
	{
		// second tier (64-slot tables, a fixed grid looping over the groups), then whatever is left one wavefront per item
		A.in_items = over1; A.in_count = C_NOVER; A.out_over = over2; A.out_count = C_NOVER2;
		if(ws->stats_on) hipLaunchKernelGGL((scan_se_kernel<true, 0, TS2>), dim3(TIER2_GRID), dim3(STHREADS), 0, stream, A);
		else hipLaunchKernelGGL((scan_se_kernel<false, 0, TS2>), dim3(TIER2_GRID), dim3(STHREADS), 0, stream, A);
		A.in_items = over2; A.in_count = C_NOVER2;
		hipLaunchKernelGGL(scan_dense_kernel, dim3((unsigned) ws->dense_slots), dim3(64), 0, stream, A);
	}
	const unsigned cgrid = (unsigned) ((n + CB - 1) / CB);
	hipLaunchKernelGGL(combine_count_kernel, dim3(cgrid), dim3(CB), 0, stream, A, out->rc_flag, out->flag, out->T_off, ws->blk_sums);
	hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, ws->blk_sums, (int64_t) cgrid);
	hipLaunchKernelGGL(combine_write_kernel, dim3(cgrid), dim3(CB), 0, stream, A, out->rc_flag, out->flag, out->T_off, ws->blk_sums, out->T, out->T_cap);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

int kmahip_launch_chain_anchors(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, KmaAnk *pool, int64_t pool_cap,
                                int64_t *a_off, int32_t *a_n, uint8_t *slow, unsigned long long *cnt, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	if(n <= 0 || !p) { kmahip_set_error("bad arguments"); return KMAHIP_EINVAL; }
	int rc = ws_reserve(ws, n);
	if(rc) return rc;
	AnchorArgs A;
	ScanArgs &S = A.S;
	S.db = db->dev;
	S.n_reads = n; S.seq = reads->seq; S.seq_off = reads->seq_off; S.len = reads->len; S.N = reads->N; S.N_off = reads->N_off;
	S.M = p->rw.M; S.MM = p->rw.MM; S.U = p->rw.U; S.W1 = p->rw.W1; S.exhaustive = p->exhaustive;
	S.item_score = ws->item_score; S.item_n = ws->item_n; S.item_off = ws->item_off;
	S.pool = ws->pool; S.pool_cap = ws->pool_cap; S.counters = ws->counters; S.overflow_items = ws->overflow_items;
	S.dense = ws->dense; S.dense_slots = ws->dense_slots; S.active_items = ws->active_items;
	S.mode = 0; S.pool_sc = nullptr; S.pool_tail0 = 2 * n * INL; S.ablate = 0; S.cat_bases = 0;
	S.in_items = ws->active_items; S.in_count = C_NACT; S.out_over = ws->overflow_items; S.out_count = C_NOVER;
	A.pool = pool; A.pool_cap = pool_cap; A.a_off = a_off; A.a_n = a_n; A.slow = slow; A.cnt = cnt;
	HIP_TRY(hipMemsetAsync(ws->counters, 0, sizeof(unsigned long long), stream));
	HIP_TRY(hipMemsetAsync(ws->counters + 2, 0, (N_COUNTERS - 2) * sizeof(unsigned long long), stream));
	HIP_TRY(hipMemsetAsync(a_n, 0, (size_t) n * 2 * sizeof(int32_t), stream));
	HIP_TRY(hipMemsetAsync(slow, 0, (size_t) n, stream));
	HIP_TRY(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), stream));
	const int64_t items = 2 * n;
	hipLaunchKernelGGL((scan_prefilter_kernel<false>), dim3((unsigned) ((items + PF_ITEMS - 1) / PF_ITEMS)), dim3(THREADS), 0, stream, S);
	A.vstride = (std::min(CA_NPMAX, std::max(1, reads->max_len - (int) db->info.kmersize + 1)) + 1) | 1;
	hipLaunchKernelGGL(chain_anchor_kernel, dim3((unsigned) ((items + GROUP - 1) / GROUP)), dim3(THREADS), (size_t) GROUP * A.vstride * sizeof(uint32_t), stream, A);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

int kmahip_launch_scan_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                          kmahip_pe_recs *out, hipStream_t stream) {
	const int64_t n = reads->n_reads;
	if(n < 0 || (n & 1) || !out || !p) { kmahip_set_error("paired scan needs an even number of reads (mates interleaved)"); return KMAHIP_EINVAL; }
	const int64_t np = n / 2;
	int rc = ws_reserve(ws, n > 0 ? n : 1);
	if(rc) return rc;
	if(ws->pe_cap < ws->pool_cap) {
		(void) hipFree(ws->pool_sc); (void) hipFree(ws->ppool); (void) hipFree(ws->pe_rec);
		ws->pool_sc = nullptr; ws->ppool = nullptr; ws->pe_rec = nullptr;
		HIP_TRY(hipMalloc((void **) &ws->pool_sc, ws->pool_cap * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->ppool, 2 * ws->pool_cap * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->pe_rec, (size_t) (ws->cap_reads + 2) * (sizeof(int32_t) + sizeof(int64_t))));
		ws->pe_cap = ws->pool_cap;
	}
	ScanArgs A;
	A.db = db->dev;
	A.n_reads = n; A.seq = reads->seq; A.seq_off = reads->seq_off; A.len = reads->len; A.N = reads->N; A.N_off = reads->N_off;
	A.M = p->rw.M; A.MM = p->rw.MM; A.U = p->rw.U; A.W1 = p->rw.W1; A.exhaustive = p->exhaustive;
	A.item_score = ws->item_score; A.item_n = ws->item_n; A.item_off = ws->item_off;
	A.pool = ws->pool; A.pool_cap = ws->pool_cap; A.counters = ws->counters; A.overflow_items = ws->overflow_items;
	A.dense = ws->dense; A.dense_slots = ws->dense_slots; A.active_items = ws->active_items;
	A.ablate = 0; A.mode = 1; A.pool_sc = ws->pool_sc; A.pool_tail0 = 0; A.cat_bases = diag_cat_bases(db);
#ifdef KMAHIP_DIAG
	if(const char *e = getenv("KMAHIP_ABLATE_SCAN")) A.ablate = atoi(e);
#endif
	HIP_TRY(hipMemsetAsync(ws->counters, 0, sizeof(unsigned long long), stream));
	HIP_TRY(hipMemsetAsync(ws->counters + 2, 0, (N_COUNTERS - 2) * sizeof(unsigned long long), stream));
	if(n == 0) { HIP_TRY(hipMemsetAsync(out->R_off, 0, sizeof(int64_t), stream)); return KMAHIP_OK; }
	hipEvent_t ev0 = nullptr, ev1 = nullptr, evp = nullptr, evq = nullptr;
	if(ws->timing_on) {
		HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventCreate(&evp)); HIP_TRY(hipEventCreate(&evq));
		HIP_TRY(hipEventRecord(evp, stream));
	}
	if(ws->stats_on) hipLaunchKernelGGL((scan_prefilter_kernel<true>), dim3((unsigned) ((2 * n + PF_ITEMS - 1) / PF_ITEMS)), dim3(THREADS), 0, stream, A);
	else hipLaunchKernelGGL((scan_prefilter_kernel<false>), dim3((unsigned) ((2 * n + PF_ITEMS - 1) / PF_ITEMS)), dim3(THREADS), 0, stream, A);
	if(ws->timing_on) { HIP_TRY(hipEventRecord(evq, stream)); HIP_TRY(hipEventRecord(ev0, stream)); }
	const unsigned grid = (unsigned) ((2 * n + SG - 1) / SG);
	int64_t *over1 = ws->overflow_items, *over2 = ws->overflow_items + 2 * ws->cap_reads;
	A.in_items = ws->active_items; A.in_count = C_NACT; A.out_over = over1; A.out_count = C_NOVER;
	if(ws->stats_on) hipLaunchKernelGGL((scan_se_kernel<true, 1, TS1>), dim3(grid), dim3(STHREADS), 0, stream, A);
	else hipLaunchKernelGGL((scan_se_kernel<false, 1, TS1>), dim3(grid), dim3(STHREADS), 0, stream, A);
	if(ws->timing_on) {
		HIP_TRY(hipEventRecord(ev1, stream));
		if(!ws->events) ws->events = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
		ws->events->push_back({ev0, ev1});
		if(!ws->events3) ws->events3 = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
		ws->events3->push_back({evp, evq});
	}
	A.in_items = over1; A.in_count = C_NOVER; A.out_over = over2; A.out_count = C_NOVER2;
	hipLaunchKernelGGL((scan_se_kernel<false, 1, TS2>), dim3(TIER2_GRID), dim3(STHREADS), 0, stream, A);
	A.in_items = over2; A.in_count = C_NOVER2;
	hipLaunchKernelGGL(scan_dense_kernel, dim3((unsigned) ws->dense_slots), dim3(64), 0, stream, A);
	PairArgs P;
	P.S = A; P.n_pairs = np; P.PE = p->rw.PE;
	P.ppool = ws->ppool; P.ppool_cap = 2 * ws->pool_cap;
	P.r_mate = out->mate; P.r_rc = out->rc; P.r_score = out->rc_flag; P.r_flag = out->flag;
	P.r_off = (int64_t *) ws->pe_rec; P.r_n = (int32_t *) (P.r_off + ws->cap_reads + 2);
	if((p->apm & 3) == 1) hipLaunchKernelGGL(pair_union_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, stream, P);
	else if((p->apm & 3) == 2) hipLaunchKernelGGL(pair_force_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, stream, P);
	else hipLaunchKernelGGL(pair_penalty_kernel, dim3((unsigned) ((np + 255) / 256)), dim3(256), 0, stream, P);
	const unsigned cgrid = (unsigned) ((n + CB - 1) / CB);
	hipLaunchKernelGGL(rec_count_kernel, dim3(cgrid), dim3(CB), 0, stream, P.r_n, n, ws->blk_sums);
	hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, ws->blk_sums, (int64_t) cgrid);
	hipLaunchKernelGGL(rec_write_kernel, dim3(cgrid), dim3(CB), 0, stream, P.r_n, P.r_off, ws->ppool, n, ws->blk_sums,
	                   out->R_off, out->T, out->T_cap, ws->counters);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}
