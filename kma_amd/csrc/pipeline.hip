// pipeline.hip -- kmahip_run_se: the single-end `-1t1` run of one batch in one call (host buffers in, per-template results
// out), built from the same launchers as the stage-wise entry points; what runKMA does between its input stream and the
// `.res` / consensus output (runkma.c:104-900), minus the files.
#include "kmahip_internal.h"
#include <chrono>
#include <vector>

namespace {

struct DevBlock {
	std::vector<void *> owned;
	~DevBlock() { for(void *p : owned) (void) hipFree(p); }
	template <class T> int get(size_t n, T **dst, bool zero = false) {
		void *d = nullptr;
		if(hipMalloc(&d, (n ? n : 1) * sizeof(T)) != hipSuccess) { kmahip_set_error("hipMalloc of %zu bytes failed", n * sizeof(T)); return KMAHIP_ENOMEM; }
		owned.push_back(d);
		if(zero && hipMemset(d, 0, (n ? n : 1) * sizeof(T)) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
		*dst = (T *) d;
		return KMAHIP_OK;
	}
	template <class T> int up(const T *src, size_t n, size_t pad, const T **dst) {
		T *d = nullptr;
		int rc = get(n + pad, &d);
		if(rc) return rc;
		if(pad && hipMemset(d + n, 0, pad * sizeof(T)) != hipSuccess) { kmahip_set_error("hipMemset failed"); return KMAHIP_EDEVICE; }
		if(n && hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { kmahip_set_error("hipMemcpy failed"); return KMAHIP_EDEVICE; }
		*dst = d;
		return KMAHIP_OK;
	}
};

double since(std::chrono::steady_clock::time_point &t) {
	const auto now = std::chrono::steady_clock::now();
	const double ms = std::chrono::duration<double, std::milli>(now - t).count();
	t = now;
	return ms;
}

// status word of the workspace after a synchronised stage (and the first counter, the pool / run top)
int ws_status(kmahip_ws *ws, unsigned long long *c0) {
	unsigned long long c[2];
	if(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return -1;
	if(c[1]) (void) hipMemset(ws->counters + 1, 0, sizeof(unsigned long long));
	if(c0) *c0 = c[0];
	return (int) c[1];
}

}  // namespace

extern "C" int kmahip_run_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, double evalue, int bcd,
                             int64_t max_frag, kmahip_run *out) {
	if(!db || !ws || !reads || !p || !out || !out->rows || !out->assembly.cover || !out->assembly.aln_len || !out->assembly.depth || !out->assembly.asm_len) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	const size_t D = db->info.DB_size;
	for(int i = 0; i < 6; ++i) out->ms[i] = 0;
	out->n_rows = 0;
	hipStream_t s = 0;
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	DevBlock B;
	int rc;
	auto t = std::chrono::steady_clock::now();

	// the batch, once
	kmahip_reads d = *reads;
	if((rc = B.up(reads->seq, (size_t) reads->seq_words, 2, &d.seq)) || (rc = B.up(reads->seq_off, (size_t) n + 1, 0, &d.seq_off)) ||
	   (rc = B.up(reads->len, (size_t) n, 1, &d.len)) || (rc = B.up(reads->N, (size_t) reads->N_total, 1, &d.N)) ||
	   (rc = B.up(reads->N_off, (size_t) n + 1, 0, &d.N_off))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	out->ms[0] = since(t);

	// stage 2; the candidate lists have no bound known in advance: start at 2 per read and redo with the exact size if short
	kmahip_cands c;
	if((rc = B.get((size_t) n + 1, &c.rc_flag)) || (rc = B.get((size_t) n + 1, &c.flag)) || (rc = B.get((size_t) n + 1, &c.T_off))) return rc;
	int64_t total = 0;
	c.T_cap = 2 * n + 4096;
	for(int attempt = 0;; ++attempt) {
		if((rc = B.get((size_t) c.T_cap, &c.T))) return rc;
		if((rc = kmahip_launch_scan_se(db, ws, &d, p, &c, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		const int st = ws_status(ws, nullptr);
		if(st == 1) {
			if(attempt >= 4) { kmahip_set_error("internal candidate pool exhausted"); return KMAHIP_EOVERFLOW; }
			ws->pool_scale *= 2; ws->cap_reads = 0;     // grown by the next launch
			continue;
		}
		HIP_TRY(hipMemcpy(&total, c.T_off + n, sizeof total, hipMemcpyDeviceToHost));
		if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: scan attempt %d: %.1f ms, %lld candidates (cap %lld)\n", attempt, since(t2), (long long) total, (long long) c.T_cap); }
		if(total <= c.T_cap) break;
		c.T_cap = total + 1024;
	}
	if(dbg) { auto t2 = t; fprintf(stderr, "[kmahip] run_se: stage 2 done after %.1f ms\n", since(t2)); }

	// stage 3a
	kmahip_hits h;
	if((rc = B.get((size_t) n + 1, &h.n_hits, true)) || (rc = B.get((size_t) n + 1, &h.best_score, true)) || (rc = B.get((size_t) n + 1, &h.flag, true)) ||
	   (rc = B.get((size_t) n + 1, &h.rc, true)) || (rc = B.get((size_t) total + 1, &h.tmpl, true)) || (rc = B.get((size_t) total + 1, &h.score, true)) ||
	   (rc = B.get((size_t) total + 1, &h.start, true)) || (rc = B.get((size_t) total + 1, &h.end, true)) ||
	   (rc = B.get(D, &h.alignment_scores, true)) || (rc = B.get(D, &h.uniq_alignment_scores, true))) return rc;
	if(n && (rc = kmahip_launch_align_se(db, ws, &d, &c, p, &h, s))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	if(ws_status(ws, nullptr) == 3) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); return KMAHIP_EOVERFLOW; }
	out->ms[1] = since(t);

	// stage 3b + the `.res` statistics (host arithmetic on one u64 per template)
	kmahip_conclave cc;
	if((rc = B.get((size_t) n + 1, &cc.tmpl, true)) || (rc = B.get((size_t) n + 1, &cc.start, true)) || (rc = B.get((size_t) n + 1, &cc.end, true)) ||
	   (rc = B.get(D, &cc.w_scores, true))) return rc;
	cc.fragment_counts = nullptr; cc.read_counts = nullptr; cc.depth = nullptr;
	if(n && (rc = kmahip_conclave_se_dev(db, ws, &d, &c, &h, &cc, s))) return rc;
	std::vector<uint64_t> w(D);
	HIP_TRY(hipMemcpy(w.data(), cc.w_scores, D * 8, hipMemcpyDeviceToHost));
	if((rc = kmahip_res_rows(db, w.data(), evalue, p->scoreT, out->rows, out->rows_cap, &out->n_rows))) return rc;
	std::vector<uint8_t> ok(D + 8, 0);
	for(int64_t r = 0; r < out->n_rows; ++r) ok[(size_t) out->rows[r].template_id] = (uint8_t) out->rows[r].significant;
	const uint8_t *d_ok = nullptr;
	if((rc = B.up(ok.data(), D + 8, 0, &d_ok))) return rc;
	out->ms[2] = since(t);

	// stage 3c per read; the run pool is sized for a handful of runs per read and grown on demand
	kmahip_traces tr;
	if((rc = B.get((size_t) 10 * n + 10, &tr.stats)) || (rc = B.get((size_t) n + 1, &tr.ops_off)) || (rc = B.get((size_t) n + 1, &tr.n_ops))) return rc;
	tr.ops_cap = 6 * n + (1 << 20);
	for(int attempt = 0; n; ++attempt) {
		if((rc = B.get((size_t) tr.ops_cap, &tr.ops))) return rc;
		if((rc = kmahip_launch_trace(db, ws, &d, h.rc, cc.tmpl, d_ok, p, &tr, s))) return rc;
		HIP_TRY(hipStreamSynchronize(s));
		unsigned long long used = 0;
		const int st = ws_status(ws, &used);
		if(st == 2 || (int64_t) used > tr.ops_cap) {
			if(attempt >= 2) { kmahip_set_error("alignment run pool: %llu runs needed", used); return KMAHIP_EOVERFLOW; }
			tr.ops_cap = (int64_t) used + (1 << 20);
			continue;
		}
		if(st) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %d)", st); return KMAHIP_EDEVICE; }
		break;
	}
	out->ms[3] = since(t);

	// stage 3c per template
	if(n) {
		if((rc = kmahip_assemble_dev(db, ws, &d, h.rc, cc.tmpl, &tr, max_frag, bcd, evalue, &out->assembly))) return rc;
	} else for(size_t i = 0; i < D; ++i) { out->assembly.cover[i] = 0; out->assembly.aln_len[i] = 0; out->assembly.depth[i] = 0; out->assembly.asm_len[i] = 0; }
	out->ms[4] = since(t);     // (pile-up + copy-back + consensus: kmahip_assemble_dev prints the split with KMAHIP_DEBUG_TIMING)

	// per-read columns for the text writers
	if(n) {
		if(out->tmpl) HIP_TRY(hipMemcpy(out->tmpl, cc.tmpl, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(out->n_hits) HIP_TRY(hipMemcpy(out->n_hits, h.n_hits, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(out->rc) HIP_TRY(hipMemcpy(out->rc, h.rc, (size_t) n * 4, hipMemcpyDeviceToHost));
		if(out->trace_stats) HIP_TRY(hipMemcpy(out->trace_stats, tr.stats, (size_t) n * 40, hipMemcpyDeviceToHost));
	}
	out->ms[5] = since(t);
	return KMAHIP_OK;
}
