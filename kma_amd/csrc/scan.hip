// scan.hip -- stage 2 of KMA on gfx950: k-mer extraction, probe of the template
// k-mer table in HBM and per-read candidate-template scoring for `-1t1`
// single-end reads.  Behaviour restated from save_kmers (savekmers.c:2442-3065),
// getBestMatch (savekmers.c:273-294) and hashMap_getGlobal (hashmapkma.c:149-178);
// the structure is new:
//
//   work item  = one (read, strand); 64 items (32 reads) per 256-thread workgroup
//   phase 0    = prefilter: every k-th k-mer of every N-free segment is probed by
//                4 lanes per item; a strand with no hit is dropped (:2477-2495)
//   phase 1    = all 256 lanes probe every k-mer start of the active items; the
//                value-set offset (or MISS) of each position goes to an LDS tile
//                laid out [position][item] (conflict-free for both phases)
//   phase 2    = one wavefront walks the tile, lane = item, running the
//                sequential run-length score machine (:2511-2706) against a small
//                per-item candidate table in LDS (id, score, last-hit position)
//   overflow   = items whose candidate set exceeds the LDS table are redone by
//                scan_dense_kernel with DB_size-wide tables in HBM
//   combine    = per read strand pick / tie merge (:3037-3062) and CSR output
//
// Reverse strand k-mers are the reverse complement of forward k-mers read at the
// mirrored position, so only the forward 2-bit words are ever staged.
#include "kmahip_internal.h"
#include <cstdlib>

namespace {

constexpr int ITEMS = 64;
constexpr int THREADS = 256;
constexpr int CHUNK = 136;
constexpr int SW = 7;                 // staged u64 words per item and pass
constexpr int TCAP = 24;              // LDS candidate-table capacity per item
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
	int32_t *dense;
	int64_t dense_slots;
};

enum { C_POOL = 0, C_STATUS = 1, C_NOVER = 2, C_PROBES = 3, C_VALS = 4, C_ACTIVE = 5 };

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
	uint64_t x = lo << ip;
	if(ip) x |= hi >> (64 - ip);
	return x >> (64 - 2 * k);
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

template <bool STATS>
__global__ __launch_bounds__(THREADS) void scan_se_kernel(const ScanArgs A) {
	__shared__ uint32_t vi_buf[CHUNK * ITEMS];
	__shared__ uint64_t w_lds[ITEMS * SW];
	__shared__ uint32_t t_id[TCAP * ITEMS];
	__shared__ int32_t t_score[TCAP * ITEMS];
	__shared__ int32_t t_ext[TCAP * ITEMS];
	__shared__ uint8_t t_cur[TCAP * ITEMS];
	__shared__ int32_t s_len[ITEMS], s_nN[ITEMS], s_wbase[ITEMS];
	__shared__ int64_t s_soff[ITEMS], s_noff[ITEMS];
	__shared__ uint32_t s_active[2];
	__shared__ int32_t s_maxnpos;
	__shared__ uint32_t s_stats[2];

	const DevDB &db = A.db;
	const int tid = threadIdx.x;
	const int k = (int) db.kmersize;
	const int64_t item0 = (int64_t) blockIdx.x * ITEMS;

	if(tid < ITEMS) {
		const int64_t r = (item0 + tid) >> 1;
		int L = 0, nN = 0; int64_t so = 0, no = 0;
		if(r < A.n_reads) {
			L = A.len[r]; so = A.seq_off[r]; no = A.N_off[r]; nN = (int) (A.N_off[r + 1] - no);
		}
		s_len[tid] = L; s_nN[tid] = nN; s_soff[tid] = so; s_noff[tid] = no;
	}
	if(tid < 2) { s_active[tid] = 0; s_stats[tid] = 0; }
	if(tid == 0) s_maxnpos = 0;
	__syncthreads();

	// ---- phase 0: prefilter --------------------------------------------
	{
		const int a = tid & 63, slot = tid >> 6;
		const int L = s_len[a], nN = s_nN[a], strand = a & 1, npos = L - k + 1;
		bool hit = false;
		uint32_t nprobe = 0;
		if(npos > 0) {
			const uint64_t *rs = A.seq + s_soff[a];
#ifdef KMAHIP_DIAG
			if(A.exhaustive || (A.ablate & 4)) {
#else
			if(A.exhaustive) {
#endif
				hit = (a & 1) == 0 || A.exhaustive;
			} else if(nN == 0) {
				for(int j = slot * k; j < npos; j += 4 * k) {
					const int q = strand ? (L - k - j) : j;
					const int w = q >> 5;
					uint64_t km = kmer_from(rs[w], rs[w + 1], q, k);
					if(strand) km = revcomp_kmer(km, k);
					++nprobe;
					if(probe(db, (uint32_t) km) != MISS) { hit = true; break; }
				}
			} else if(slot == 0) {
				// rare: walk the N-free segments exactly like savekmers.c:2483-2495
				const int32_t *Nl = A.N + s_noff[a];
				int j = 0;
				for(int i = 1; i <= nN + 1 && !hit; ++i) {
					const int segend = n_strand(Nl, nN, L, strand, i);
					for(; j < segend - k + 1 && !hit; j += k) {
						const int q = strand ? (L - k - j) : j;
						const int w = q >> 5;
						uint64_t km = kmer_from(rs[w], rs[w + 1], q, k);
						if(strand) km = revcomp_kmer(km, k);
						++nprobe;
						if(probe(db, (uint32_t) km) != MISS) hit = true;
					}
					j = segend + 1;
				}
			}
		}
		if(hit) {
			atomicOr(&s_active[a >> 5], 1u << (a & 31));
			atomicMax(&s_maxnpos, npos);
		}
		if(STATS && nprobe) atomicAdd(&s_stats[0], nprobe);
	}
	__syncthreads();
	const uint64_t active = ((uint64_t) s_active[1] << 32) | s_active[0];
	const int maxnpos = s_maxnpos;

	// ---- per-item machine state: 16 items per wave, lanes 0-15 of each of the 4 waves --
	// (the other lanes only probe). Spreading the 64 machines over all four waves lets
	// their value-list loads overlap; inside a wave the lanes advance in "rounds": every
	// lane first walks its LDS column (no global access) up to its next value-set change,
	// then all lanes handle one change each, so the serial depth is the number of set
	// changes per item (~30 for a 150 bp read), not the number of k-mer positions.
	uint32_t last = NONE;
	int gaps = 0, HIT = 0, acc = 0, nlist = 0, ncur = 0, hits = 0;
	bool overflow = false;
	const int lane = tid & 63;
	const int a_own = ((tid >> 6) << 4) | (lane & 15);
	const bool owner = lane < 16;
	const bool my_active = owner && ((active >> a_own) & 1ull);

	// one template of a newly opened value set (savekmers.c:2584-2655)
	auto open_template = [&](const int a, const uint32_t t) {
		int e = -1;
		for(int x = 0; x < nlist; ++x) if(t_id[x * ITEMS + a] == t) { e = x; break; }
		if(e >= 0) {
			t_score[e * ITEMS + a] += bridge(HIT - t_ext[e * ITEMS + a], k, A.M, A.MM, A.U, A.W1);
		} else {
			if(nlist == TCAP) { overflow = true; return; }
			e = nlist++;
			t_id[e * ITEMS + a] = t;
			t_score[e * ITEMS + a] = k * A.M;
		}
		t_cur[ncur * ITEMS + a] = (uint8_t) e;
		++ncur;
	};

	for(int c0 = 0; c0 < maxnpos; c0 += CHUNK) {
		// stage forward words covering this pass
		for(int idx = tid; idx < ITEMS * SW; idx += THREADS) {
			const int a = idx / SW, w = idx - a * SW;
			uint64_t v = 0;
			if((active >> a) & 1ull) {
				const int L = s_len[a], npos = L - k + 1;
				int lo = c0;
				if(a & 1) { const int jmax = min(c0 + CHUNK, npos) - 1; lo = L - k - jmax; }
				if(lo < 0) lo = 0;
				const int wb = lo >> 5;
				if(w == 0) s_wbase[a] = wb;
				const int words = (L + 31) >> 5;
				if(wb + w < words) v = A.seq[s_soff[a] + wb + w];
			}
			w_lds[idx] = v;
		}
		__syncthreads();
		// probe every k-mer start of the pass
		uint32_t nprobe = 0;
		for(int idx = tid; idx < ITEMS * CHUNK; idx += THREADS) {
			const int a = idx & 63, jj = idx >> 6;
			if(!((active >> a) & 1ull)) continue;
			const int L = s_len[a], p = c0 + jj;
			if(p >= L - k + 1) continue;
			const int q = (a & 1) ? (L - k - p) : p;
			uint32_t vi = MISS;
			const int nN = s_nN[a];
			if(nN == 0 || !window_has_N(A.N + s_noff[a], nN, q, k)) {
				const int w = (q >> 5) - s_wbase[a];
				uint64_t km = kmer_from(w_lds[a * SW + w], w_lds[a * SW + w + 1], q, k);
				if(a & 1) km = revcomp_kmer(km, k);
#ifdef KMAHIP_DIAG
				if(A.ablate & 2) vi = (uint32_t) (km & 1023u); else
#endif
				vi = probe(db, (uint32_t) km);
				++nprobe;
			}
			vi_buf[jj * ITEMS + a] = vi;
		}
		if(STATS && nprobe) atomicAdd(&s_stats[0], nprobe);
		__syncthreads();
		// score machines, in rounds of one value-set change per lane
		{
			const int a = a_own;
#ifdef KMAHIP_DIAG
			const bool run = my_active && !overflow && !(A.ablate & 1);
#else
			const bool run = my_active && !overflow;
#endif
			const int jend = run ? min(CHUNK, s_len[a] - k + 1 - c0) : 0;
			int jj = 0;
			for(;;) {
				uint32_t vi = MISS;
				bool pending = false;
				while(jj < jend) {
					vi = vi_buf[jj * ITEMS + a];
					if(vi == MISS) { ++gaps; ++jj; continue; }
					if(vi == last) { acc += bridge(gaps, k, A.M, A.MM, A.U, A.W1); HIT = c0 + jj; gaps = 0; ++hits; ++jj; continue; }
					pending = true;
					break;
				}
				if(!__any(pending)) break;
				if(pending) {
					const int p = c0 + jj;
					// fetch the new list first: count + up to 7 ids in flight together
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
					// close the old set (savekmers.c:2575-2582)
					for(int c = 0; c < ncur; ++c) {
						const int e = t_cur[c * ITEMS + a];
						t_score[e * ITEMS + a] += acc;
						t_ext[e * ITEMS + a] = HIT;
					}
					HIT = p - 1;
					ncur = 0;
					if(STATS) atomicAdd(&s_stats[1], cnt + 1u);
#pragma unroll
					for(int i = 0; i < 7; ++i) if((uint32_t) i < cnt && !overflow) open_template(a, el[i]);
					for(uint32_t i = 8; i <= cnt && !overflow; ++i) open_template(a, value_at(db, vi, (int) i));
					if(overflow) { jj = jend; }
					else { last = vi; acc = 0; HIT = p; gaps = 0; ++hits; ++jj; }
				}
			}
		}
		__syncthreads();
	}

	// ---- finish items -------------------------------------------------------
	if(owner) {
		const int a = a_own;
		const int64_t item = item0 + a;
		if((item >> 1) < A.n_reads) {
			int best = 0, nb = 0;
			int64_t off = 0;
			if(my_active && overflow) {
				const unsigned long long slot = atomicAdd(&A.counters[C_NOVER], 1ull);
				A.overflow_items[slot] = item;
				nb = -1;
			} else if(my_active && hits) {
				for(int c = 0; c < ncur; ++c) t_score[t_cur[c * ITEMS + a] * ITEMS + a] += acc;
				for(int e = 0; e < nlist; ++e) {
					const int s = max(0, t_score[e * ITEMS + a]);
					if(s > best) { best = s; nb = 1; } else if(s == best) ++nb;
				}
				if(best > 0) {
					off = (int64_t) atomicAdd(&A.counters[C_POOL], (unsigned long long) nb);
					if(off + nb <= A.pool_cap) {
						int w = 0;
						for(int e = 0; e < nlist; ++e) {
							if(max(0, t_score[e * ITEMS + a]) == best) A.pool[off + w++] = (int32_t) t_id[e * ITEMS + a];
						}
					} else {
						atomicMax(&A.counters[C_STATUS], 1ull);
					}
				} else {
					nb = 0;
				}
			}
			A.item_score[item] = best;
			A.item_n[item] = nb;
			A.item_off[item] = off;
		}
	}
	if(STATS) {
		__syncthreads();
		if(tid == 0) {
			atomicAdd(&A.counters[C_PROBES], (unsigned long long) s_stats[0]);
			atomicAdd(&A.counters[C_VALS], (unsigned long long) s_stats[1]);
			atomicAdd(&A.counters[C_ACTIVE], (unsigned long long) __popcll(active));
		}
	}
}

// Overflow path: one lane per item, DB_size-wide score / last-hit / list arrays
// in HBM (the reference's own per-thread layout, savekmers.c:134-150).
__global__ __launch_bounds__(64) void scan_dense_kernel(const ScanArgs A) {
	const DevDB &db = A.db;
	const int k = (int) db.kmersize;
	const int64_t n_over = (int64_t) A.counters[C_NOVER];
	const int64_t slot = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(slot >= A.dense_slots) return;
	const int64_t D = db.DB_size;
	int32_t *score = A.dense + slot * 3 * D;
	int32_t *ext = score + D;
	int32_t *list = ext + D;   // list[0..nlist); membership: ext[t] high bit trick avoided -> score[t] = INT_MIN marks "absent"
	const int64_t *items = A.overflow_items;
	for(int64_t oi = slot; oi < n_over; oi += A.dense_slots) {
		const int64_t item = items[oi];
		const int64_t r = item >> 1;
		const int strand = (int) (item & 1);
		const int L = A.len[r], npos = L - k + 1;
		const uint64_t *rs = A.seq + A.seq_off[r];
		const int32_t *Nl = A.N + A.N_off[r];
		const int nN = (int) (A.N_off[r + 1] - A.N_off[r]);
		uint32_t last = NONE;
		int gaps = 0, HIT = 0, acc = 0, nlist = 0, hits = 0;
		for(int p = 0; p < npos; ++p) {
			const int q = strand ? (L - k - p) : p;
			uint32_t vi = MISS;
			if(nN == 0 || !window_has_N(Nl, nN, q, k)) {
				const int w = q >> 5;
				uint64_t km = kmer_from(rs[w], rs[w + 1], q, k);
				if(strand) km = revcomp_kmer(km, k);
				vi = probe(db, (uint32_t) km);
			}
			if(vi == MISS) { ++gaps; continue; }
			if(vi == last) {
				acc += bridge(gaps, k, A.M, A.MM, A.U, A.W1);
			} else {
				if(last != NONE) {
					const int c = (int) value_at(db, last, 0);
					for(int i = 1; i <= c; ++i) { const uint32_t t = value_at(db, last, i); score[t] += acc; ext[t] = HIT; }
				}
				HIT = p - 1;
				const int cnt = (int) value_at(db, vi, 0);
				for(int i = 1; i <= cnt; ++i) {
					const uint32_t t = value_at(db, vi, i);
					if(ext[t] != -1) {
						score[t] += bridge(HIT - ext[t], k, A.M, A.MM, A.U, A.W1);
					} else {
						score[t] = k * A.M;
						ext[t] = 0;
						list[nlist++] = (int32_t) t;
					}
				}
				last = vi;
				acc = 0;
			}
			HIT = p; gaps = 0; ++hits;
		}
		int best = 0, nb = 0;
		int64_t off = 0;
		if(hits) {
			const int c = (int) value_at(db, last, 0);
			for(int i = 1; i <= c; ++i) score[value_at(db, last, i)] += acc;
			for(int e = 0; e < nlist; ++e) {
				const int s = max(0, score[list[e]]);
				if(s > best) { best = s; nb = 1; } else if(s == best) ++nb;
			}
			if(best > 0) {
				off = (int64_t) atomicAdd(&A.counters[C_POOL], (unsigned long long) nb);
				if(off + nb <= A.pool_cap) {
					int w = 0;
					for(int e = 0; e < nlist; ++e) if(max(0, score[list[e]]) == best) A.pool[off + w++] = list[e];
				} else {
					atomicMax(&A.counters[C_STATUS], 1ull);
				}
			} else {
				nb = 0;
			}
			for(int e = 0; e < nlist; ++e) { score[list[e]] = 0; ext[list[e]] = -1; }
		}
		A.item_score[item] = best;
		A.item_n[item] = nb;
		A.item_off[item] = off;
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

} // namespace

static int ws_reserve(kmahip_ws *ws, int64_t n_reads) {
	kmahip_db *db = ws->db;
	if(n_reads > ws->cap_reads) {
		(void) hipFree(ws->item_score); (void) hipFree(ws->item_n); (void) hipFree(ws->item_off);
		(void) hipFree(ws->pool); (void) hipFree(ws->overflow_items); (void) hipFree(ws->blk_sums);
		ws->item_score = ws->item_n = nullptr; ws->item_off = nullptr; ws->pool = nullptr;
		ws->overflow_items = nullptr; ws->blk_sums = nullptr;
		const int64_t cap = n_reads + n_reads / 8 + 1024;
		HIP_TRY(hipMalloc((void **) &ws->item_score, cap * 2 * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->item_n, cap * 2 * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->item_off, cap * 2 * sizeof(int64_t)));
		ws->pool_cap = cap * 16;
		HIP_TRY(hipMalloc((void **) &ws->pool, ws->pool_cap * sizeof(int32_t)));
		HIP_TRY(hipMalloc((void **) &ws->overflow_items, cap * 2 * sizeof(int64_t)));
		ws->blk_cap = (cap + CB - 1) / CB + 1;
		HIP_TRY(hipMalloc((void **) &ws->blk_sums, ws->blk_cap * sizeof(int64_t)));
		ws->cap_reads = cap;
	}
	if(!ws->counters) { HIP_TRY(hipMalloc((void **) &ws->counters, 8 * sizeof(unsigned long long))); HIP_TRY(hipMemset(ws->counters, 0, 8 * sizeof(unsigned long long))); }
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
	A.dense = ws->dense; A.dense_slots = ws->dense_slots;
	A.ablate = 0;
#ifdef KMAHIP_DIAG
	if(const char *e = getenv("KMAHIP_ABLATE_SCAN")) A.ablate = atoi(e);
#endif
	// word 1 (status) is sticky until kmahip_ws_status reads it
	HIP_TRY(hipMemsetAsync(ws->counters, 0, sizeof(unsigned long long), stream));
	HIP_TRY(hipMemsetAsync(ws->counters + 2, 0, 6 * sizeof(unsigned long long), stream));
	if(n == 0) {
		HIP_TRY(hipMemsetAsync(out->T_off, 0, sizeof(int64_t), stream));
		return KMAHIP_OK;
	}
	const int64_t items = 2 * n;
	const unsigned grid = (unsigned) ((items + ITEMS - 1) / ITEMS);
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	if(ws->timing_on) {
		HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
		HIP_TRY(hipEventRecord(ev0, stream));
	}
	if(ws->stats_on) hipLaunchKernelGGL(scan_se_kernel<true>, dim3(grid), dim3(THREADS), 0, stream, A);
	else hipLaunchKernelGGL(scan_se_kernel<false>, dim3(grid), dim3(THREADS), 0, stream, A);
	if(ws->timing_on) {
		HIP_TRY(hipEventRecord(ev1, stream));
		if(!ws->events) ws->events = new std::vector<std::pair<hipEvent_t, hipEvent_t>>();
		ws->events->push_back({ev0, ev1});
	}
	{
		const unsigned dgrid = (unsigned) ((ws->dense_slots + 63) / 64);
		hipLaunchKernelGGL(scan_dense_kernel, dim3(dgrid), dim3(64), 0, stream, A);
	}
	const unsigned cgrid = (unsigned) ((n + CB - 1) / CB);
	hipLaunchKernelGGL(combine_count_kernel, dim3(cgrid), dim3(CB), 0, stream, A, out->rc_flag, out->flag, out->T_off, ws->blk_sums);
	hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, ws->blk_sums, (int64_t) cgrid);
	hipLaunchKernelGGL(combine_write_kernel, dim3(cgrid), dim3(CB), 0, stream, A, out->rc_flag, out->flag, out->T_off, ws->blk_sums, out->T, out->T_cap);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}
