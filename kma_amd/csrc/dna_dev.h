// dna_dev.h -- device helpers shared by the stage-3 kernels (align.hip, longtrace.hip): views of 2-bit packed reads in
// either orientation, 32-base windows, the per-template position index lookup (hashMapCCI_get semantics,
// hashmapcci.c:95-124) and the substitution model of chain.c.
#pragma once
#include "kmahip_internal.h"

namespace {

struct Aln { int score, len, pos, match, tGaps, qGaps; };

struct QView {
	const uint64_t *w;
	const int32_t *N;
	int L, nN, rc;
	// seed bounds in the coordinates of this view (the query bounds a default-mode S2 record carries, qseqs.c:41-56): the seed
	// search starts at b0 and the last N-free stretch ends at b1 (KMA_score align.c:534-540, KMA :249-254). Whole read by default.
	int b0 = 0, b1 = 0x7fffffff;
};
__device__ __forceinline__ int qb1(const QView &q) { return q.b1 < q.L ? q.b1 : q.L; }
// bounds given for the read as stored -> this view (reverse complemented views count from the other end, alnfrags.c:1113-1127)
__device__ __forceinline__ void q_set_bounds(QView &q, const int32_t *q_start, const int32_t *q_end, int64_t rd) {
	if(!q_start) return;
	const int s = q_start[rd], e = q_end[rd];
	if(q.rc) { q.b0 = q.L - e; q.b1 = q.L - s; } else { q.b0 = s; q.b1 = e; }
}

__device__ __forceinline__ int q2(const QView &q, int i) {
	const int p = q.rc ? q.L - 1 - i : i;
	const int b = (int) ((q.w[p >> 5] >> (62 - ((p & 31) << 1))) & 3ull);
	return q.rc ? 3 - b : b;
}

__device__ __forceinline__ bool q_is_N(const QView &q, int i) {
	if(q.nN == 0) return false;
	const int p = q.rc ? q.L - 1 - i : i;
	int lo = 0, hi = q.nN;
	while(lo < hi) { const int mid = (lo + hi) >> 1; if(q.N[mid] < p) lo = mid + 1; else hi = mid; }
	return lo < q.nN && q.N[lo] == p;
}

// byte code of the oriented query: 0-3, 4 = N (unCompDNA, compdna.c:178-203)
__device__ __forceinline__ int qn(const QView &q, int i) { return q_is_N(q, i) ? 4 : q2(q, i); }

// i-th (1-based) oriented N position; i > nN -> L (the sentinel alnFragsSE appends)
__device__ __forceinline__ int qN_at(const QView &q, int i) {
	if(i > q.nN) return q.L;
	return q.rc ? (q.L - 1 - q.N[q.nN - i]) : q.N[i - 1];
}

__device__ __forceinline__ uint32_t q_kmer(const QView &q, int j, int k) {
	const int p = q.rc ? q.L - k - j : j;
	const int ip = (p & 31) << 1, w = p >> 5;
	uint64_t x = q.w[w] << ip;
	if(ip) x |= q.w[w + 1] >> (64 - ip);
	x >>= (64 - 2 * k);
	if(q.rc) {
		x = ~x;
		x = __brevll(x);
		x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
		x >>= (64 - 2 * k);
	}
	return (uint32_t) x;
}

__device__ __forceinline__ int tn(const uint64_t *ts, int pos) {
	return (int) ((ts[pos >> 5] >> (62 - ((pos & 31) << 1))) & 3ull);
}

__device__ __forceinline__ uint64_t revcomp64(uint64_t x) {
	x = __brevll(~x);
	return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}

// 32 bases starting at base `pos` of a 2-bit word array (MSB first). Always reads word and word+1 (every array carries a
// pad word), as one 16-byte access and without a branch: a conditional second load would wait for the first one.
__device__ __forceinline__ uint64_t win2(const uint64_t *w, int pos) {
	const int ip = (pos & 31) << 1;
	const uint64_t *p = w + (pos >> 5);
	const uint64_t w0 = p[0], w1 = p[1];
	return (w0 << ip) | ((w1 >> 1) >> (63 - ip));
}

// 32 bases of the ORIENTED read starting at oriented position i (N packed as A;
// bases past the read end are garbage)
__device__ __forceinline__ uint64_t qwin(const QView &q, int i) {
	if(!q.rc) return win2(q.w, i);
	const int s = q.L - 32 - i;           // forward window that mirrors [i, i+32)
	if(s >= 0) return revcomp64(win2(q.w, s));
	return revcomp64(q.w[0] >> ((-s) << 1));
}

// the k-mers at oriented positions j and j + 1 from ONE 32-base window (k <= 16 here, so both fit): one 16-byte access
// instead of up to four word loads and two reverse complements
__device__ __forceinline__ void q_kmer2(const QView &q, int j, int k, uint32_t &km1, uint32_t &km2) {
	const uint64_t w = qwin(q, j);
	km1 = (uint32_t) (w >> (64 - 2 * k));
	km2 = (uint32_t) ((w << 2) >> (64 - 2 * k));
}

// Query codes of a DP problem without a global load per cell: a cached 32-base window of the oriented
// read, reloaded when the column index leaves it (columns are walked in descending order).
struct QCursor {
	uint64_t w;
	int blk;      // window covers problem columns [32*blk, 32*blk + 32)
	__device__ __forceinline__ int code(const QView &q, int q_s, int n) {
		const int b = n >> 5;
		if(b != blk) { blk = b; w = qwin(q, q_s + (b << 5)); }
		if(q.nN && q_is_N(q, q_s + n)) return 4;
		return (int) ((w >> (62 - ((n & 31) << 1))) & 3ull);
	}
};

// hashMapCCI_get semantics (hashmapcci.c:95-124): 0 absent, +pos unique, negative = duplicated. A probe takes the aligned PAIR of
// slots its slot lies in (16 bytes: tables start at even slots and have an even number of them), so a chain of slots is walked two per
// round trip; with the tables at most a third full (db.hip) few lookups need a second one.
__device__ __forceinline__ bool tpos_probe(const uint4 &e, uint32_t sl, uint32_t km, int &v) {      // true: decided (v = value or 0)
	if(!(sl & 1u)) {
		if(e.y == 0) { v = 0; return true; }
		if(e.x == km) { v = (int) e.y; return true; }
	}
	if(e.w == 0) { v = 0; return true; }
	if(e.z == km) { v = (int) e.w; return true; }
	return false;
}
__device__ __forceinline__ int tpos_get(const DevDB &db, int t, uint32_t km) {
	if(km == 0) return 0;
	const uint32_t sh = db.tpos_shift[t];
	const uint2 *tab = db.tpos_slots + db.tpos_off[t];
	const uint32_t msk = (1u << (32 - sh)) - 1u;
	uint32_t sl = (km * 0x9E3779B1u) >> sh;
	for(;;) {
		const uint4 e = *(const uint4 *) (tab + (sl & ~1u));
		int v;
		if(tpos_probe(e, sl, km, v)) return v;
		sl = ((sl | 1u) + 1u) & msk;
	}
}

// two lookups whose first table gathers travel together
__device__ __forceinline__ void tpos_get2(const uint2 *tab, uint32_t sh, uint32_t km1, uint32_t km2, bool want2, int &v1, int &v2);
__device__ __forceinline__ void tpos_get2(const DevDB &db, int t, uint32_t km1, uint32_t km2, bool want2, int &v1, int &v2) {
	tpos_get2(db.tpos_slots + db.tpos_off[t], db.tpos_shift[t], km1, km2, want2, v1, v2);
}
__device__ __forceinline__ void tpos_get2(const uint2 *tab, uint32_t sh, uint32_t km1, uint32_t km2, bool want2, int &v1, int &v2) {
	const uint32_t msk = (1u << (32 - sh)) - 1u;
	uint32_t s1 = (km1 * 0x9E3779B1u) >> sh, s2 = (km2 * 0x9E3779B1u) >> sh;
	uint4 e1 = *(const uint4 *) (tab + (s1 & ~1u)), e2 = *(const uint4 *) (tab + ((want2 ? s2 : s1) & ~1u));
	v1 = 0; v2 = 0;
	if(km1) while(!tpos_probe(e1, s1, km1, v1)) { s1 = ((s1 | 1u) + 1u) & msk; e1 = *(const uint4 *) (tab + s1); }
	if(want2 && km2) while(!tpos_probe(e2, s2, km2, v2)) { s2 = ((s2 | 1u) + 1u) & msk; e2 = *(const uint4 *) (tab + s2); }
}

// heuristic substitution model of chain.c (end / link / start terms)
__device__ __forceinline__ int mism_score(int span, int k, int M, int MM) {
	int Ms, MMs;
	if(span == 2) { MMs = 2; Ms = 0; }
	else {
		MMs = span / k + (span % k ? 1 : 0);
		MMs = max(2, MMs);
		Ms = min(min(span - MMs, k), MMs);
	}
	return Ms * M + MMs * MM;
}


} // namespace
