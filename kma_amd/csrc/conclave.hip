// conclave.hip -- stage 3b of KMA on gfx950: ConClave, the choice of ONE template per mapped read from the
// (globally summed) alignment_scores / uniq_alignment_scores vectors, and the per-template totals the `.res`
// rows start from. Behaviour restated from runConClave (conclave.c:43-215, the default `-ConClave 1`) and
// runKMA's row statistics (runkma.c:608-613, 765-783; p_chisqr / fastp, stdstat.c:36-147).
// The reference streams frag_raw records through one thread; here every record is one lane: the choice reads only
// the two read-only vectors, and the per-template sums are order-free u64 atomics.
#include "kmahip_internal.h"
#include <cmath>

namespace {

// -lc (kma.c:694-701) points ConClavePtr to runConClave_lc: one setting per process, like the reference's function pointer
static int g_conclave_lc = 0;
extern "C" int kmahip_set_conclave_lc(int on) { g_conclave_lc = on != 0; return KMAHIP_OK; }

struct CCArgs {
	int64_t n_slots;             // SE: reads; PE: 2 * pairs (record slots in stream order)
	int pe;                      // 0: one slot per single-end read; 1: record slots of kmahip_align_pe_dev; 2: explicit records
	const int32_t *len;          // read lengths (PE: mates interleaved)
	const int32_t *len2;         // mode 2: length of the mate carried by a pair record
	const int32_t *mate;         // PE: which mate a record slot carries (-1: none)
	const int32_t *pe_kind;      // PE: per pair (kmahip_align_pe_dev)
	const int64_t *off;          // SE: T_off; PE: R_off
	const int32_t *n_hits, *best_score;
	const int32_t *h_tmpl, *h_start, *h_end;
	const uint64_t *as, *us;     // alignment_scores, uniq_alignment_scores
	const int32_t *tlen;
	int lc;                      // -lc: runConClave_lc's order of the tests
	int32_t *o_tmpl, *o_start, *o_end;
	unsigned long long *w_scores, *depth;
	uint32_t *frag_counts, *read_counts;
	// a record with an empty list takes the first listed hit of the last record BEFORE it that had any (below); for the first such
	// records of a read shard that is a record of an earlier shard: what it left behind (0, 0, 0 at the start of the stream)
	int carry_t, carry_s, carry_e;
	int32_t *last_out;           // conclave_last_kernel: {valid, tmpl, start, end} of this shard's last record with a list
};

struct Rec { int n, score, q_len, q_len2; int64_t o; };

// The frag_raw record (if any) that stage 3a wrote for slot s (updatescores.c:283-295, 360-388, 470-488).
// score < 0: a pair (the mate follows in the same record, conclave.c:171).
__device__ bool record_at(const CCArgs &A, int64_t s, Rec &r) {
	r.n = 0; r.score = 0; r.q_len = 0; r.q_len2 = 0; r.o = 0;
	if(A.pe == 2) {
		r.n = abs(A.n_hits[s]); r.score = A.best_score[s]; r.o = A.off[s]; r.q_len = A.len[s]; r.q_len2 = A.len2 ? A.len2[s] : 0;
		return r.n > 0 || r.score != 0;
	}
	if(!A.pe) {
		r.n = A.n_hits[s];
		if(r.n <= 0) return false;
		r.score = A.best_score[s]; r.o = A.off[s]; r.q_len = A.len[s];
		return true;
	}
	const int64_t p0 = s & ~1ll;
	const int x = (int) (s & 1), kind = A.pe_kind[p0 >> 1];
	auto mate_len = [&](int64_t slot) { const int m = A.mate[slot]; return m < 0 ? 0 : A.len[p0 + m]; };
	switch(kind) {
	case 0:
		r.n = A.n_hits[s];
		if(r.n <= 0) return false;
		r.score = A.best_score[s]; r.o = A.off[s]; r.q_len = mate_len(s);
		return true;
	case 1:      // proper pair: ONE record, written even when its hit list came out empty (updatescores.c:402-417)
		if(!x) return false;
		r.n = A.n_hits[s]; r.score = -A.best_score[s]; r.o = A.off[s]; r.q_len = mate_len(p0); r.q_len2 = mate_len(p0 + 1);
		return true;
	case 2:      // unmated: two single records, both lists in the second slot's slice
		r.n = A.n_hits[s]; r.score = A.best_score[s]; r.o = A.off[p0 + 1] + (x ? A.n_hits[p0] : 0); r.q_len = mate_len(s);
		return r.n > 0 || r.score != 0;
	case 3: case 4:
		if(x != (kind == 4)) return false;
		r.n = A.n_hits[s]; r.score = A.best_score[s]; r.o = A.off[p0 + 1]; r.q_len = mate_len(s);
		return r.n > 0;
	}
	return false;
}

__global__ __launch_bounds__(256) void conclave_kernel(const CCArgs A) {
	const int64_t s = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
	if(s >= A.n_slots) return;
	Rec r;
	int tt = 0, st = 0, en = 0;
	const bool written = record_at(A, s, r);
	if(written) {
		if(r.n > 1) {
			// conclave.c:59-133. The running best score / uniq count live in `int`s there and are compared with the
			// unsigned long vector entries: truncation and sign extension are part of the behaviour.
			int best_tmpl = -1, best_read_score = 0, best_num = 0, pick = -1;
			double best_score = 0;
			for(int i = 0; i < r.n; ++i) {
				const int x = A.h_tmpl[r.o + i], t = abs(x);
				const uint64_t a = A.as[t], u = A.us[t];
				const double sc = 1.0 * (double) a / (double) A.tlen[t];
				bool take = false;
				// (-lc, runConClave_lc conclave.c:215-385: the score per template base decides before the score itself)
				const bool first_gt = A.lc ? sc > best_score : a > (uint64_t) (int64_t) best_read_score;
				const bool first_eq = A.lc ? sc == best_score : a == (uint64_t) (int64_t) best_read_score;
				const bool second_gt = A.lc ? a > (uint64_t) (int64_t) best_read_score : sc > best_score;
				const bool second_eq = A.lc ? a == (uint64_t) (int64_t) best_read_score : sc == best_score;
				if(first_gt) take = true;
				else if(first_eq) {
					if(second_gt) take = true;
					else if(second_eq) {
						if(u > (uint64_t) (int64_t) best_num) take = true;
						else if(u == (uint64_t) (int64_t) best_num && t < abs(best_tmpl)) take = true;
					}
				}
				if(take) { pick = i; best_tmpl = x; best_read_score = (int) a; best_score = sc; best_num = (int) u; }
			}
			if(pick >= 0) { tt = A.h_tmpl[r.o + pick]; st = A.h_start[r.o + pick]; en = A.h_end[r.o + pick]; }
		} else if(r.n == 1) {
			tt = A.h_tmpl[r.o]; st = A.h_start[r.o]; en = A.h_end[r.o];
		} else {
			// empty list: runConClave reads zero entries and uses element 0 of its buffers (conclave.c:123-127) = the
			// first listed hit of the last record in the stream that had any
			Rec q;
			tt = A.carry_t; st = A.carry_s; en = A.carry_e;
			for(int64_t s2 = s - 1; s2 >= 0; --s2) {
				if(record_at(A, s2, q) && q.n >= 1) { tt = A.h_tmpl[q.o]; st = A.h_start[q.o]; en = A.h_end[q.o]; break; }
			}
		}
	}
	A.o_tmpl[s] = tt; A.o_start[s] = st; A.o_end[s] = en;
	const int t = abs(tt);
	if(!written || t == 0) return;
	atomicAdd(&A.w_scores[t], (unsigned long long) abs(r.score));
	if(A.frag_counts) atomicAdd(&A.frag_counts[t], 1u);
	if(A.read_counts) atomicAdd(&A.read_counts[t], r.score < 0 ? 2u : 1u);
	if(A.depth) atomicAdd(&A.depth[t], (unsigned long long) (r.q_len + (r.score < 0 ? r.q_len2 : 0)));
}

// what the records of the NEXT shard inherit: the first listed hit of this shard's last record that has a list
__global__ void conclave_last_kernel(const CCArgs A) {
	Rec q;
	A.last_out[0] = 0;
	for(int64_t s2 = A.n_slots - 1; s2 >= 0; --s2) {
		if(record_at(A, s2, q) && q.n >= 1) { A.last_out[0] = 1; A.last_out[1] = A.h_tmpl[q.o]; A.last_out[2] = A.h_start[q.o]; A.last_out[3] = A.h_end[q.o]; return; }
	}
}

} // namespace

static int launch_conclave(kmahip_db *db, const CCArgs &A0, const kmahip_hits *hits, kmahip_conclave *out, hipStream_t stream) {
	if(!hits->n_hits || !hits->best_score || !hits->tmpl || !hits->start || !hits->end || !hits->alignment_scores ||
	   !hits->uniq_alignment_scores || !out->tmpl || !out->start || !out->end || !out->w_scores) {
		kmahip_set_error("null argument"); return KMAHIP_EINVAL;
	}
	if(!db->dev.tlen) { kmahip_set_error("index has no .length.b: stage 3b unavailable"); return KMAHIP_EINVAL; }
	CCArgs A = A0;
	A.n_hits = hits->n_hits; A.best_score = hits->best_score; A.h_tmpl = hits->tmpl; A.h_start = hits->start; A.h_end = hits->end;
	A.as = hits->alignment_scores; A.us = hits->uniq_alignment_scores; A.tlen = db->dev.tlen; A.lc = g_conclave_lc;
	A.o_tmpl = out->tmpl; A.o_start = out->start; A.o_end = out->end;
	A.w_scores = (unsigned long long *) out->w_scores; A.depth = (unsigned long long *) out->depth;
	A.frag_counts = out->fragment_counts; A.read_counts = out->read_counts;
	if(A.n_slots == 0) return KMAHIP_OK;
	hipLaunchKernelGGL(conclave_kernel, dim3((unsigned) ((A.n_slots + 255) / 256)), dim3(256), 0, stream, A);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

// the reference's `cmp` (stdstat.c:23-35): how the p-value test and the score test of a template combine -- or (default), and
// (`-and`), always true (`-oa`); one setting per process, like the reference's function pointer
static int g_cmp_mode = 0;
int kmahip_cmp(bool t, bool q) { return g_cmp_mode == 0 ? (t || q) : (g_cmp_mode == 1 ? (t && q) : 1); }
extern "C" int kmahip_set_cmp(int mode) {
	if(mode < 0 || mode > 2) { kmahip_set_error("cmp mode %d (0 = or, 1 = and, 2 = true)", mode); return KMAHIP_EINVAL; }
	g_cmp_mode = mode;
	return KMAHIP_OK;
}

extern "C" int kmahip_conclave_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                                      const kmahip_hits *hits, kmahip_conclave *out, void *stream) {
	(void) ws;
	if(!db || !reads || !cands || !hits || !out || !cands->T_off || !reads->len || reads->n_reads < 0) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	CCArgs A{};
	A.n_slots = reads->n_reads; A.pe = 0; A.len = reads->len; A.off = cands->T_off;
	return launch_conclave(db, A, hits, out, (hipStream_t) stream);
}

extern "C" int kmahip_conclave_pe_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                                      const kmahip_hits *hits, const int32_t *pe_kind, kmahip_conclave *out, void *stream) {
	(void) ws;
	if(!db || !reads || !recs || !hits || !out || !pe_kind || !recs->R_off || !recs->mate || !reads->len || reads->n_reads < 0 || (reads->n_reads & 1)) {
		kmahip_set_error("null argument"); return KMAHIP_EINVAL;
	}
	CCArgs A{};
	A.n_slots = reads->n_reads; A.pe = 1; A.len = reads->len; A.off = recs->R_off; A.mate = recs->mate; A.pe_kind = pe_kind;
	return launch_conclave(db, A, hits, out, (hipStream_t) stream);
}

// explicit records with everything in HBM (the paired run builds them on the device: two slots per unit of the stream, an
// unused slot = a record with n_hits 0 and score 0, which ConClave passes over)
// for a read-sharded paired run (pipeline.hip): the records of one shard with what the shard before left behind for records with
// an empty list (carry: tmpl, start, end) ...
int kmahip_conclave_records_carry(kmahip_db *db, int64_t n_records, const int32_t *q_len, const int32_t *q_len2, const int64_t *off, const kmahip_hits *hits,
                                  kmahip_conclave *out, const int32_t carry[3], hipStream_t stream) {
	if(!db || !q_len || !off || !hits || !out || n_records < 0 || !carry) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	CCArgs A{};
	A.n_slots = n_records; A.pe = 2; A.len = q_len; A.len2 = q_len2; A.off = off;
	A.carry_t = carry[0]; A.carry_s = carry[1]; A.carry_e = carry[2];
	return launch_conclave(db, A, hits, out, stream);
}
// ... and what this shard leaves behind: d_last[4] (device) = {1, tmpl, start, end} of its last record with a list, {0, ...} if none
int kmahip_conclave_records_last(kmahip_db *db, int64_t n_records, const int32_t *q_len, const int32_t *q_len2, const int64_t *off, const kmahip_hits *hits,
                                 int32_t *d_last, hipStream_t stream) {
	if(!db || !q_len || !off || !hits || !d_last || n_records < 0) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	CCArgs A{};
	A.n_slots = n_records; A.pe = 2; A.len = q_len; A.len2 = q_len2; A.off = off;
	A.n_hits = hits->n_hits; A.best_score = hits->best_score; A.h_tmpl = hits->tmpl; A.h_start = hits->start; A.h_end = hits->end;
	A.last_out = d_last;
	hipLaunchKernelGGL(conclave_last_kernel, dim3(1), dim3(1), 0, stream, A);
	HIP_TRY(hipGetLastError());
	return KMAHIP_OK;
}

extern "C" int kmahip_conclave_records_dev(kmahip_db *db, kmahip_ws *ws, int64_t n_records, const int32_t *q_len, const int32_t *q_len2,
                                           const int64_t *off, const kmahip_hits *hits, kmahip_conclave *out, void *stream) {
	(void) ws;
	if(!db || !q_len || !off || !hits || !out || n_records < 0) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	CCArgs A{};
	A.n_slots = n_records; A.pe = 2; A.len = q_len; A.len2 = q_len2; A.off = off;
	return launch_conclave(db, A, hits, out, (hipStream_t) stream);
}

// ---- host-buffer forms: stage everything, run, copy back (PCIe inclusive; glue for callers that keep stage 3a's
// results on the host) ------------------------------------------------------------------------------------------
namespace {
struct DevBuf {
	std::vector<void *> owned;
	~DevBuf() { for(void *p : owned) (void) hipFree(p); }
	template <class T> int up(const T *src, size_t n, T **dst) {
		void *d = nullptr;
		if(hipMalloc(&d, (n ? n : 1) * sizeof(T)) != hipSuccess) { kmahip_set_error("hipMalloc failed"); return KMAHIP_EDEVICE; }
		owned.push_back(d);
		if(n && src && hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
		if(n && !src && hipMemset(d, 0, n * sizeof(T)) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
		*dst = (T *) d;
		return KMAHIP_OK;
	}
};
template <class T> int down(T *dst, const T *src, size_t n) {
	if(n && dst && hipMemcpy(dst, src, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) { kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
	return KMAHIP_OK;
}
} // namespace

static int conclave_host(kmahip_db *db, int pe, int64_t n, const int32_t *len, const int32_t *len2, const int32_t *mate,
                         const int32_t *pe_kind, const int64_t *off, const kmahip_hits *hits, kmahip_conclave *out) {
	if(n < 0 || (pe == 1 && (n & 1))) { kmahip_set_error("bad record count"); return KMAHIP_EINVAL; }
	if(!hits->n_hits || !hits->best_score || !hits->tmpl || !hits->start || !hits->end || !hits->alignment_scores ||
	   !hits->uniq_alignment_scores || !out->tmpl || !out->start || !out->end || !out->w_scores) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size, nn = (size_t) n, total = n ? (size_t) off[n] : 0;
	DevBuf B;
	CCArgs A{};
	kmahip_hits dh{};
	kmahip_conclave dc{};
	int32_t *d_len, *d_len2 = nullptr, *d_mate = nullptr, *d_kind = nullptr;
	int64_t *d_off;
	int rc;
	if((rc = B.up(len, nn, &d_len)) || (rc = B.up(off, nn + 1, &d_off)) ||
	   (rc = B.up(hits->n_hits, nn, &dh.n_hits)) || (rc = B.up(hits->best_score, nn, &dh.best_score)) ||
	   (rc = B.up(hits->tmpl, total, &dh.tmpl)) || (rc = B.up(hits->start, total, &dh.start)) || (rc = B.up(hits->end, total, &dh.end)) ||
	   (rc = B.up(hits->alignment_scores, D, &dh.alignment_scores)) || (rc = B.up(hits->uniq_alignment_scores, D, &dh.uniq_alignment_scores)) ||
	   (rc = B.up((const int32_t *) nullptr, nn, &dc.tmpl)) || (rc = B.up((const int32_t *) nullptr, nn, &dc.start)) ||
	   (rc = B.up((const int32_t *) nullptr, nn, &dc.end)) || (rc = B.up(out->w_scores, D, &dc.w_scores))) return rc;
	if(pe == 1 && ((rc = B.up(mate, nn, &d_mate)) || (rc = B.up(pe_kind, nn / 2, &d_kind)))) return rc;
	if(pe == 2 && len2 && (rc = B.up(len2, nn, &d_len2))) return rc;
	if(out->fragment_counts && (rc = B.up(out->fragment_counts, D, &dc.fragment_counts))) return rc;
	if(out->read_counts && (rc = B.up(out->read_counts, D, &dc.read_counts))) return rc;
	if(out->depth && (rc = B.up(out->depth, D, &dc.depth))) return rc;
	A.n_slots = n; A.pe = pe; A.len = d_len; A.len2 = d_len2; A.off = d_off; A.mate = d_mate; A.pe_kind = d_kind;
	if((rc = launch_conclave(db, A, &dh, &dc, 0))) return rc;
	HIP_TRY(hipDeviceSynchronize());
	if((rc = down(out->tmpl, dc.tmpl, nn)) || (rc = down(out->start, dc.start, nn)) || (rc = down(out->end, dc.end, nn)) ||
	   (rc = down(out->w_scores, dc.w_scores, D)) || (rc = down(out->fragment_counts, dc.fragment_counts, D)) ||
	   (rc = down(out->read_counts, dc.read_counts, D)) || (rc = down(out->depth, dc.depth, D))) return rc;
	return KMAHIP_OK;
}

extern "C" int kmahip_conclave_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                                  const kmahip_hits *hits, kmahip_conclave *out) {
	(void) ws;
	if(!db || !reads || !cands || !hits || !out || !cands->T_off || !reads->len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return conclave_host(db, 0, reads->n_reads, reads->len, nullptr, nullptr, nullptr, cands->T_off, hits, out);
}

extern "C" int kmahip_conclave_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                                  const kmahip_hits *hits, const int32_t *pe_kind, kmahip_conclave *out) {
	(void) ws;
	if(!db || !reads || !recs || !hits || !out || !pe_kind || !recs->R_off || !recs->mate || !reads->len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return conclave_host(db, 1, reads->n_reads, reads->len, nullptr, recs->mate, pe_kind, recs->R_off, hits, out);
}

extern "C" int kmahip_conclave_records(kmahip_db *db, kmahip_ws *ws, int64_t n_records, const int32_t *q_len, const int32_t *q_len2,
                                       const int64_t *off, const kmahip_hits *hits, kmahip_conclave *out) {
	(void) ws;
	if(!db || !q_len || !off || !hits || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return conclave_host(db, 2, n_records, q_len, q_len2, nullptr, nullptr, off, hits, out);
}

// ---- `.res` row statistics (host arithmetic; the reference computes them in long double, runkma.c:141) ----------

// fastp, stdstat.c:36-134: chi-square (1 d.o.f.) quantile -> p-value by table
static double chi2_table(long double q) {
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
	return 1.00 - chi2_table(-1 * q);
}

// p_chisqr, stdstat.c:136-147
static double p_chisqr(long double q) {
	if(q < 0) return 1e-26;
	if(q > 49) return chi2_table(q);
	return 1 - 1.772453850 * erf(sqrt((double) (0.5 * q))) / tgamma(0.5);
}

// runkma.c:578-583, 608-613, 765-783 for every template with a ConClave score, in template order
double kmahip_p_chisqr(long double q) { return p_chisqr(q); }

extern "C" int kmahip_res_rows(const kmahip_db *db, const uint64_t *w_scores, double evalue, double scoreT,
                               kmahip_res_row *rows, int64_t cap, int64_t *n_rows) {
	if(!db || !w_scores || !n_rows || (cap > 0 && !rows)) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int D = (int) db->info.DB_size;
	if((int) db->h_tlen.size() < D) { kmahip_set_error("index has no .length.b"); return KMAHIP_EINVAL; }
	long unsigned Nhits = 0, tot = 0;
	for(int i = D - 1; i > 0; --i) { tot += (long unsigned) db->h_tlen[i]; Nhits += w_scores[i]; }
	Nhits = Nhits ? Nhits : 1;
	int64_t n = 0;
	for(int t = 1; t < D; ++t) {
		if(!(w_scores[t] > 0)) continue;
		if(n < cap) {
			const long read_score = (long) w_scores[t];
			const int t_len = db->h_tlen[t];
			long double expected = t_len, q_value;
			const long unsigned rest = tot - t_len;
			expected /= (1 < rest ? rest : 1);
			expected *= (Nhits - read_score);
			if(0 < expected) {
				q_value = read_score - expected;
				q_value /= (expected + read_score);
				q_value *= (read_score - expected);
			} else q_value = read_score;
			const double p_value = p_chisqr(q_value);
			kmahip_res_row &r = rows[n];
			r.template_id = t; r.template_length = t_len; r.score = w_scores[t];
			r.expected = (unsigned) expected; r.q_value = (double) q_value; r.p_value = p_value;
			r.significant = kmahip_cmp(p_value <= evalue && read_score > expected, read_score >= scoreT * t_len);
		}
		++n;
	}
	*n_rows = n;
	if(n > cap) { kmahip_set_error("row capacity %lld < %lld", (long long) cap, (long long) n); return KMAHIP_EOVERFLOW; }
	return KMAHIP_OK;
}
