// api.hip -- extern "C" entry points of libkmahip.so (see include/kmahip.h).
#include "kmahip_internal.h"
void kmahip_devcache_flush();      // pipeline.hip
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <vector>

extern "C" int kmahip_ws_create(kmahip_db *db, kmahip_ws **out) {
	if(!db || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	kmahip_ws *ws = new kmahip_ws();
	memset(ws, 0, sizeof *ws);
	ws->db = db;
	*out = ws;
	return KMAHIP_OK;
}

extern "C" int kmahip_ws_set_pe_chain(kmahip_ws *ws, const kmahip_chain_params *cp) {
	if(!ws) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	ws->pe_chain_on = cp != nullptr;
	if(cp) ws->pe_chain = *cp;
	return KMAHIP_OK;
}

extern "C" void kmahip_ws_destroy(kmahip_ws *ws) {
	if(!ws) return;
	(void) hipFree(ws->item_score); (void) hipFree(ws->item_n); (void) hipFree(ws->item_off);
	(void) hipFree(ws->pool); (void) hipFree(ws->counters); (void) hipFree(ws->overflow_items); (void) hipFree(ws->active_items);
	(void) hipFree(ws->dense); (void) hipFree(ws->blk_sums);
	for(int i = 0; i < 8; ++i) (void) hipFree(ws->stage[i]);
	for(auto *ev : {ws->events, ws->events2, ws->events3, ws->events4}) {
		if(!ev) continue;
		for(auto &e : *ev) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
		delete ev;
	}
	(void) hipFree(ws->a_s32); (void) hipFree(ws->a_s64); (void) hipFree(ws->a_task); (void) hipFree(ws->a_priv); (void) hipFree(ws->a_xq);
	if(ws->a_side) { (void) hipStreamDestroy(ws->a_side); (void) hipEventDestroy(ws->a_ev[0]); (void) hipEventDestroy(ws->a_ev[1]); }
	(void) hipFree(ws->t_s32); (void) hipFree(ws->t_E); (void) hipFree(ws->t_queue);
	(void) hipFree(ws->p_counts); (void) hipFree(ws->p_chain); (void) hipFree(ws->p_seg); (void) hipFree(ws->p_vals);
	(void) hipFree(ws->p_nodes); (void) hipFree(ws->p_keys); (void) hipFree(ws->p_rank);
	(void) hipFree(ws->pool_sc); (void) hipFree(ws->ppool); (void) hipFree(ws->pe_rec);
	kmahip_devcache_flush();          // (the large blocks the runs keep between calls: pipeline.hip)
	for(int i = 0; i < 20; ++i) { if(i == 17) (void) hipHostFree(ws->lt_buf[i]); else (void) hipFree(ws->lt_buf[i]); }      // (17: pinned host memory, longtrace.hip)
	delete ws;
}

extern "C" int kmahip_scan_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                                  const kmahip_params *p, kmahip_cands *out, void *stream) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_scan_se(db, ws, reads, p, out, (hipStream_t) stream);
}

extern "C" int kmahip_ws_status(kmahip_ws *ws, void *stream) {
	if(!ws) return KMAHIP_EINVAL;
	if(!ws->counters) return KMAHIP_OK;   // nothing launched yet
	unsigned long long c[8];
	HIP_TRY(hipMemcpyAsync(c, ws->counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t) stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
	if(c[1]) {
		HIP_TRY(hipMemset(ws->counters + 1, 0, sizeof(unsigned long long)));
		switch(c[1]) {
			case 1:
				// the library's own candidate pool (cap_reads * 16 * pool_scale ints), not a caller capacity: grown by the next launch
				ws->pool_scale *= 2; ws->cap_reads = 0;
				kmahip_set_error("internal candidate pool exhausted: the pool has been doubled, repeat the call");
				break;
			case 2: kmahip_set_error("output capacity (T_cap / ops_cap) too small: the offsets array holds the needed size; stage 3a was skipped"); break;
			case 3: case 16: {
				// more MEMs against one template than the scratch has slots per (read, template) pair (a read full of repeats): the
				// capacity goes up fourfold with the next launch, like the candidate pool
				const int cur = ws->mem_scale > 0 ? ws->mem_scale : 1;
				if(cur < 64) { ws->mem_scale = cur * 4; kmahip_set_error("seed (MEM) capacity per read/template pair exceeded: it has been raised fourfold, repeat the %s call (with the score vectors zeroed again)", c[1] == 3 ? "align" : "trace"); }
				else kmahip_set_error("seed (MEM) capacity per read/template pair exceeded");
				break;
			}
			default: kmahip_set_error("a read needs more scratch than the workspace holds (status %llu)", c[1]); break;
		}
		return KMAHIP_EOVERFLOW;
	}
	return KMAHIP_OK;
}

extern "C" int kmahip_scan_set_stats(kmahip_ws *ws, int on) {
	if(!ws) return KMAHIP_EINVAL;
	ws->stats_on = on;
	return KMAHIP_OK;
}

extern "C" int kmahip_scan_get_stats(kmahip_ws *ws, kmahip_scan_stats *st, void *stream) {
	if(!ws || !st || !ws->counters) return KMAHIP_EINVAL;
	unsigned long long c[KMAHIP_N_COUNTERS];
	HIP_TRY(hipMemcpyAsync(c, ws->counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t) stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
	st->probes = c[3]; st->value_elems = c[4]; st->active_strands = c[5]; st->hash_probes = c[6]; st->prefilter_probes = c[9];
	return KMAHIP_OK;
}

extern "C" int kmahip_align_get_stats(kmahip_ws *ws, kmahip_align_stats *st, void *stream) {
	if(!ws || !st || !ws->counters) return KMAHIP_EINVAL;
	unsigned long long c[8];
	HIP_TRY(hipMemcpyAsync(c, ws->counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t) stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
	st->lookups = c[3]; st->mem_bases = c[4]; st->dp_cells = c[5]; st->tasks = c[6];
	return KMAHIP_OK;
}

static int stage_reserve(kmahip_ws *ws, int slot, size_t bytes) {
	if(bytes == 0) bytes = 8;
	if(ws->stage_bytes[slot] >= bytes) return KMAHIP_OK;
	(void) hipFree(ws->stage[slot]);
	ws->stage[slot] = nullptr; ws->stage_bytes[slot] = 0;
	bytes += bytes / 4;
	HIP_TRY(hipMalloc(&ws->stage[slot], bytes));
	ws->stage_bytes[slot] = bytes;
	return KMAHIP_OK;
}

// host buffers in / out: reads are staged once; stage 3a optionally follows on the same staged batch
static int run_host(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                    kmahip_cands *out, kmahip_hits *hits) {
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	int rc;
	// stage inputs: 0 seq, 1 seq_off, 2 len, 3 N, 4 N_off; outputs: 5 rc_flag+flag, 6 T_off, 7 T
	if((rc = stage_reserve(ws, 0, (size_t) (reads->seq_words + 1) * 8)) || (rc = stage_reserve(ws, 1, (size_t) (n + 1) * 8)) ||
	   (rc = stage_reserve(ws, 2, (size_t) n * 4)) || (rc = stage_reserve(ws, 3, (size_t) reads->N_total * 4)) ||
	   (rc = stage_reserve(ws, 4, (size_t) (n + 1) * 8)) || (rc = stage_reserve(ws, 5, (size_t) n * 8)) ||
	   (rc = stage_reserve(ws, 6, (size_t) (n + 1) * 8)) || (rc = stage_reserve(ws, 7, (size_t) out->T_cap * 4))) return rc;
	hipStream_t s = 0;
	// one zero pad word after the last read keeps the word+1 access in bounds
	HIP_TRY(hipMemsetAsync((char *) ws->stage[0] + (size_t) reads->seq_words * 8, 0, 8, s));
	if(reads->seq_words) HIP_TRY(hipMemcpyAsync(ws->stage[0], reads->seq, (size_t) reads->seq_words * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[1], reads->seq_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	if(n) HIP_TRY(hipMemcpyAsync(ws->stage[2], reads->len, (size_t) n * 4, hipMemcpyHostToDevice, s));
	if(reads->N_total) HIP_TRY(hipMemcpyAsync(ws->stage[3], reads->N, (size_t) reads->N_total * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[4], reads->N_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	d.seq = (const uint64_t *) ws->stage[0]; d.seq_off = (const int64_t *) ws->stage[1]; d.len = (const int32_t *) ws->stage[2];
	d.N = (const int32_t *) ws->stage[3]; d.N_off = (const int64_t *) ws->stage[4];
	kmahip_cands o;
	o.rc_flag = (int32_t *) ws->stage[5]; o.flag = o.rc_flag + n; o.T_off = (int64_t *) ws->stage[6];
	o.T = (int32_t *) ws->stage[7]; o.T_cap = out->T_cap;
	if((rc = kmahip_launch_scan_se(db, ws, &d, p, &o, s))) return rc;
	if(n) {
		HIP_TRY(hipMemcpyAsync(out->rc_flag, o.rc_flag, (size_t) n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(out->flag, o.flag, (size_t) n * 4, hipMemcpyDeviceToHost, s));
	}
	HIP_TRY(hipMemcpyAsync(out->T_off, o.T_off, (size_t) (n + 1) * 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	unsigned long long c[8];
	HIP_TRY(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost));
	if(c[1]) HIP_TRY(hipMemset(ws->counters + 1, 0, sizeof(unsigned long long)));
	if(c[1] == 1) {
		// internal candidate pool too small: grow and let the caller retry
		ws->pool_scale *= 2; ws->cap_reads = 0;
		kmahip_set_error("internal candidate pool exhausted (retry grows it)");
		return KMAHIP_EOVERFLOW;
	}
	const int64_t total = out->T_off[n];
	if(total > out->T_cap) { kmahip_set_error("T_cap %lld too small, need %lld", (long long) out->T_cap, (long long) total); return KMAHIP_EOVERFLOW; }
	if(total) HIP_TRY(hipMemcpy(out->T, o.T, (size_t) total * 4, hipMemcpyDeviceToHost));
	if(!hits || n == 0) return KMAHIP_OK;

	// stage 3a on the staged batch; device outputs in one block
	const size_t D = db->info.DB_size;
	const size_t hb = (size_t) n * 16 + (size_t) (total + 1) * 16 + 2 * D * 8 + 64;
	void *dh = nullptr;
	HIP_TRY(hipMalloc(&dh, hb));
	HIP_TRY(hipMemsetAsync(dh, 0, hb, s));
	kmahip_hits h;
	uint64_t *u = (uint64_t *) dh;
	h.alignment_scores = u; h.uniq_alignment_scores = u + D;
	int32_t *ip = (int32_t *) (u + 2 * D);
	h.n_hits = ip; h.best_score = ip + n; h.flag = ip + 2 * n;
	h.tmpl = ip + 3 * n; h.score = h.tmpl + total + 1; h.start = h.score + total + 1; h.end = h.start + total + 1;
	h.rc = h.end + total + 1;
	rc = kmahip_launch_align_se(db, ws, &d, &o, p, &h, s);
	if(!rc) {
		hipError_t e = hipStreamSynchronize(s);
		if(e != hipSuccess) { kmahip_set_error("align kernels failed: %s", hipGetErrorString(e)); rc = KMAHIP_EDEVICE; }
	}
	if(!rc) {
		(void) hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost);
		if(c[1]) (void) hipMemset(ws->counters + 1, 0, sizeof(unsigned long long));
		if(c[1] == 3) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); rc = KMAHIP_EOVERFLOW; }
	}
	if(!rc) {
		(void) hipMemcpy(hits->n_hits, h.n_hits, (size_t) n * 4, hipMemcpyDeviceToHost);
		(void) hipMemcpy(hits->best_score, h.best_score, (size_t) n * 4, hipMemcpyDeviceToHost);
		(void) hipMemcpy(hits->flag, h.flag, (size_t) n * 4, hipMemcpyDeviceToHost);
		if(hits->rc) (void) hipMemcpy(hits->rc, h.rc, (size_t) n * 4, hipMemcpyDeviceToHost);
		if(total) {
			(void) hipMemcpy(hits->tmpl, h.tmpl, (size_t) total * 4, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hits->score, h.score, (size_t) total * 4, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hits->start, h.start, (size_t) total * 4, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hits->end, h.end, (size_t) total * 4, hipMemcpyDeviceToHost);
		}
		// ADD into the caller's accumulators
		std::vector<uint64_t> acc(2 * D);
		(void) hipMemcpy(acc.data(), u, 2 * D * 8, hipMemcpyDeviceToHost);
		if(hits->alignment_scores) for(size_t i = 0; i < D; ++i) hits->alignment_scores[i] += acc[i];
		if(hits->uniq_alignment_scores) for(size_t i = 0; i < D; ++i) hits->uniq_alignment_scores[i] += acc[D + i];
		hipError_t e = hipGetLastError();
		if(e != hipSuccess) { kmahip_set_error("copy back failed: %s", hipGetErrorString(e)); rc = KMAHIP_EDEVICE; }
	}
	(void) hipFree(dh);
	return rc;
}

extern "C" int kmahip_scan_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                              const kmahip_params *p, kmahip_cands *out) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return run_host(db, ws, reads, p, out, nullptr);
}

extern "C" int kmahip_map_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                             kmahip_cands *cands_out, kmahip_hits *hits_out) {
	if(!db || !ws || !reads || !p || !cands_out || !hits_out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return run_host(db, ws, reads, p, cands_out, hits_out);
}

extern "C" int kmahip_ws_set_timing(kmahip_ws *ws, int on) {
	if(!ws) return KMAHIP_EINVAL;
	ws->timing_on = on;
	return KMAHIP_OK;
}

extern "C" int kmahip_ws_get_timing(kmahip_ws *ws, int kernel, double *total_ms, int64_t *launches) {
	if(!ws || !total_ms || !launches || kernel < 0 || kernel > 3) return KMAHIP_EINVAL;
	*total_ms = 0.0; *launches = 0;
	auto *ev = kernel == 0 ? ws->events : kernel == 1 ? ws->events2 : kernel == 2 ? ws->events3 : ws->events4;
	if(!ev) return KMAHIP_OK;
	for(auto &e : *ev) {
		float ms = 0.f;
		HIP_TRY(hipEventSynchronize(e.second));
		HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
		*total_ms += ms; *launches += 1;
		(void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second);
	}
	ev->clear();
	return KMAHIP_OK;
}

extern "C" int kmahip_align_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_cands *cands,
                                   const kmahip_params *p, kmahip_hits *out, void *stream) {
	if(!db || !ws || !reads || !cands || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_align_se(db, ws, reads, cands, p, out, (hipStream_t) stream);
}

// RCCL is bound lazily so that a single-GPU host never needs librccl and a host that
// already loaded one (e.g. through PyTorch) keeps using that copy.
extern "C" int kmahip_allreduce_scores(void *nccl_comm, uint64_t *alignment_scores, uint64_t *uniq_alignment_scores,
                                       size_t DB_size, void *stream) {
	typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
	static allreduce_fn fn = nullptr;
	if(!nccl_comm || !alignment_scores || !uniq_alignment_scores) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(!fn) {
		fn = (allreduce_fn) dlsym(RTLD_DEFAULT, "ncclAllReduce");
		if(!fn) {
			void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
			if(!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
			if(h) fn = (allreduce_fn) dlsym(h, "ncclAllReduce");
		}
		if(!fn) { kmahip_set_error("RCCL (librccl.so) not found"); return KMAHIP_EDEVICE; }
	}
	// (ncclUint64, ncclSum: rccl.h at build time)
	int rc = fn(alignment_scores, alignment_scores, DB_size, ncclUint64, ncclSum, nccl_comm, (hipStream_t) stream);
	if(!rc) rc = fn(uniq_alignment_scores, uniq_alignment_scores, DB_size, ncclUint64, ncclSum, nccl_comm, (hipStream_t) stream);
	if(rc) { kmahip_set_error("ncclAllReduce failed with code %d", rc); return KMAHIP_EDEVICE; }
	return KMAHIP_OK;
}

extern "C" int kmahip_scan_pe_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                                  kmahip_pe_recs *out, void *stream) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_scan_pe(db, ws, reads, p, out, (hipStream_t) stream);
}

extern "C" int kmahip_align_pe_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_pe_recs *recs,
                                   const kmahip_params *p, kmahip_hits *out, int32_t *pe_kind, void *stream) {
	if(!db || !ws || !reads || !recs || !p || !out || !pe_kind) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_align_pe(db, ws, reads, recs, p, out, pe_kind, (hipStream_t) stream);
}

static int run_host_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, kmahip_pe_recs *out,
                       kmahip_hits *hits, int32_t *pe_kind) {
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	int rc;
	if((rc = stage_reserve(ws, 0, (size_t) (reads->seq_words + 1) * 8)) || (rc = stage_reserve(ws, 1, (size_t) (n + 1) * 8)) ||
	   (rc = stage_reserve(ws, 2, (size_t) n * 4)) || (rc = stage_reserve(ws, 3, (size_t) reads->N_total * 4)) ||
	   (rc = stage_reserve(ws, 4, (size_t) (n + 1) * 8)) || (rc = stage_reserve(ws, 5, (size_t) n * 16)) ||
	   (rc = stage_reserve(ws, 6, (size_t) (n + 1) * 8)) || (rc = stage_reserve(ws, 7, (size_t) out->T_cap * 4))) return rc;
	hipStream_t s = 0;
	HIP_TRY(hipMemsetAsync((char *) ws->stage[0] + (size_t) reads->seq_words * 8, 0, 8, s));
	if(reads->seq_words) HIP_TRY(hipMemcpyAsync(ws->stage[0], reads->seq, (size_t) reads->seq_words * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[1], reads->seq_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	if(n) HIP_TRY(hipMemcpyAsync(ws->stage[2], reads->len, (size_t) n * 4, hipMemcpyHostToDevice, s));
	if(reads->N_total) HIP_TRY(hipMemcpyAsync(ws->stage[3], reads->N, (size_t) reads->N_total * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[4], reads->N_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	d.seq = (const uint64_t *) ws->stage[0]; d.seq_off = (const int64_t *) ws->stage[1]; d.len = (const int32_t *) ws->stage[2];
	d.N = (const int32_t *) ws->stage[3]; d.N_off = (const int64_t *) ws->stage[4];
	kmahip_pe_recs o;
	int32_t *ip = (int32_t *) ws->stage[5];
	o.mate = ip; o.rc = ip + n; o.rc_flag = ip + 2 * n; o.flag = ip + 3 * n;
	o.R_off = (int64_t *) ws->stage[6]; o.T = (int32_t *) ws->stage[7]; o.T_cap = out->T_cap;
	if((rc = kmahip_launch_scan_pe(db, ws, &d, p, &o, s))) return rc;
	if(n) {
		HIP_TRY(hipMemcpyAsync(out->mate, o.mate, (size_t) n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(out->rc, o.rc, (size_t) n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(out->rc_flag, o.rc_flag, (size_t) n * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(out->flag, o.flag, (size_t) n * 4, hipMemcpyDeviceToHost, s));
	}
	HIP_TRY(hipMemcpyAsync(out->R_off, o.R_off, (size_t) (n + 1) * 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	unsigned long long c[8];
	HIP_TRY(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost));
	if(c[1]) HIP_TRY(hipMemset(ws->counters + 1, 0, sizeof(unsigned long long)));
	if(c[1] == 1) { ws->pool_scale *= 2; ws->cap_reads = 0; kmahip_set_error("internal candidate pool exhausted (retry grows it)"); return KMAHIP_EOVERFLOW; }
	const int64_t total = out->R_off[n];
	if(total > out->T_cap) { kmahip_set_error("T_cap %lld too small, need %lld", (long long) out->T_cap, (long long) total); return KMAHIP_EOVERFLOW; }
	if(total) HIP_TRY(hipMemcpy(out->T, o.T, (size_t) total * 4, hipMemcpyDeviceToHost));
	if(!hits || n == 0) return KMAHIP_OK;
	// stage 3a on the staged batch
	const size_t D = db->info.DB_size;
	const size_t hb = (size_t) n * 20 + (size_t) (total + 1) * 16 + 2 * D * 8 + 64;
	void *dh = nullptr;
	HIP_TRY(hipMalloc(&dh, hb));
	HIP_TRY(hipMemsetAsync(dh, 0, hb, s));
	kmahip_hits h;
	uint64_t *u = (uint64_t *) dh;
	h.alignment_scores = u; h.uniq_alignment_scores = u + D;
	int32_t *hp = (int32_t *) (u + 2 * D);
	h.n_hits = hp; h.best_score = hp + n; h.flag = hp + 2 * n;
	int32_t *dkind = hp + 3 * n;
	h.tmpl = hp + 4 * n; h.score = h.tmpl + total + 1; h.start = h.score + total + 1; h.end = h.start + total + 1;
	h.rc = h.end + total + 1;
	kmahip_reads d2 = d;
	rc = kmahip_launch_align_pe(db, ws, &d2, &o, p, &h, dkind, s);
	if(!rc) {
		hipError_t e = hipStreamSynchronize(s);
		if(e != hipSuccess) { kmahip_set_error("align kernels failed: %s", hipGetErrorString(e)); rc = KMAHIP_EDEVICE; }
	}
	if(!rc) {
		(void) hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost);
		if(c[1]) (void) hipMemset(ws->counters + 1, 0, sizeof(unsigned long long));
		if(c[1] == 3) { kmahip_set_error("seed (MEM) capacity per read/template pair exceeded"); rc = KMAHIP_EOVERFLOW; }
	}
	if(!rc) {
		(void) hipMemcpy(hits->n_hits, h.n_hits, (size_t) n * 4, hipMemcpyDeviceToHost);
		(void) hipMemcpy(hits->best_score, h.best_score, (size_t) n * 4, hipMemcpyDeviceToHost);
		(void) hipMemcpy(hits->flag, h.flag, (size_t) n * 4, hipMemcpyDeviceToHost);
		if(hits->rc) (void) hipMemcpy(hits->rc, h.rc, (size_t) n * 4, hipMemcpyDeviceToHost);
		(void) hipMemcpy(pe_kind, dkind, (size_t) (n / 2) * 4, hipMemcpyDeviceToHost);
		if(total) {
			(void) hipMemcpy(hits->tmpl, h.tmpl, (size_t) total * 4, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hits->score, h.score, (size_t) total * 4, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hits->start, h.start, (size_t) total * 4, hipMemcpyDeviceToHost);
			(void) hipMemcpy(hits->end, h.end, (size_t) total * 4, hipMemcpyDeviceToHost);
		}
		std::vector<uint64_t> acc(2 * D);
		(void) hipMemcpy(acc.data(), u, 2 * D * 8, hipMemcpyDeviceToHost);
		if(hits->alignment_scores) for(size_t i = 0; i < D; ++i) hits->alignment_scores[i] += acc[i];
		if(hits->uniq_alignment_scores) for(size_t i = 0; i < D; ++i) hits->uniq_alignment_scores[i] += acc[D + i];
		hipError_t e = hipGetLastError();
		if(e != hipSuccess) { kmahip_set_error("copy back failed: %s", hipGetErrorString(e)); rc = KMAHIP_EDEVICE; }
	}
	(void) hipFree(dh);
	return rc;
}

extern "C" int kmahip_scan_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p, kmahip_pe_recs *out) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return run_host_pe(db, ws, reads, p, out, nullptr, nullptr);
}

extern "C" int kmahip_map_pe(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const kmahip_params *p,
                             kmahip_pe_recs *recs_out, kmahip_hits *hits_out, int32_t *pe_kind) {
	if(!db || !ws || !reads || !p || !recs_out || !hits_out || !pe_kind) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(reads->max_len <= 0) { kmahip_set_error("kmahip_reads.max_len must be set"); return KMAHIP_EINVAL; }
	return run_host_pe(db, ws, reads, p, recs_out, hits_out, pe_kind);
}

extern "C" int kmahip_align_trace_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                                      const uint8_t *tmpl_ok, const kmahip_params *p, kmahip_traces *out, void *stream) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_trace(db, ws, reads, flag, tmpl, tmpl_ok, p, out, (hipStream_t) stream);
}

extern "C" int kmahip_align_trace(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, const int32_t *flag, const int32_t *tmpl,
                                  const uint8_t *tmpl_ok, const kmahip_params *p, kmahip_traces *out, int64_t *ops_needed) {
	if(!db || !ws || !reads || !flag || !tmpl || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0 || out->ops_cap < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(ops_needed) *ops_needed = 0;
	if(n == 0) return KMAHIP_OK;
	const size_t D = db->info.DB_size;
	int rc;
	if((rc = stage_reserve(ws, 0, (size_t) (reads->seq_words + 1) * 8)) || (rc = stage_reserve(ws, 1, (size_t) (n + 1) * 8)) ||
	   (rc = stage_reserve(ws, 2, (size_t) n * 4)) || (rc = stage_reserve(ws, 3, (size_t) reads->N_total * 4)) ||
	   (rc = stage_reserve(ws, 4, (size_t) (n + 1) * 8)) || (rc = stage_reserve(ws, 5, (size_t) n * 8 + D + 8)) ||
	   (rc = stage_reserve(ws, 6, (size_t) n * (40 + 8 + 4) + 8)) || (rc = stage_reserve(ws, 7, (size_t) (out->ops_cap + 1) * 4))) return rc;
	hipStream_t s = 0;
	HIP_TRY(hipMemsetAsync((char *) ws->stage[0] + (size_t) reads->seq_words * 8, 0, 8, s));
	if(reads->seq_words) HIP_TRY(hipMemcpyAsync(ws->stage[0], reads->seq, (size_t) reads->seq_words * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[1], reads->seq_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[2], reads->len, (size_t) n * 4, hipMemcpyHostToDevice, s));
	if(reads->N_total) HIP_TRY(hipMemcpyAsync(ws->stage[3], reads->N, (size_t) reads->N_total * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[4], reads->N_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	int32_t *d_flag = (int32_t *) ws->stage[5], *d_tmpl = d_flag + n;
	uint8_t *d_ok = tmpl_ok ? (uint8_t *) (d_tmpl + n) : nullptr;
	HIP_TRY(hipMemcpyAsync(d_flag, flag, (size_t) n * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(d_tmpl, tmpl, (size_t) n * 4, hipMemcpyHostToDevice, s));
	if(tmpl_ok) HIP_TRY(hipMemcpyAsync(d_ok, tmpl_ok, D, hipMemcpyHostToDevice, s));
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	d.seq = (const uint64_t *) ws->stage[0]; d.seq_off = (const int64_t *) ws->stage[1]; d.len = (const int32_t *) ws->stage[2];
	d.N = (const int32_t *) ws->stage[3]; d.N_off = (const int64_t *) ws->stage[4];
	kmahip_traces o;
	o.ops_off = (int64_t *) ws->stage[6]; o.stats = (int32_t *) (o.ops_off + n); o.n_ops = o.stats + 10 * n;
	o.ops = (uint32_t *) ws->stage[7]; o.ops_cap = out->ops_cap;
	if((rc = kmahip_launch_trace(db, ws, &d, d_flag, d_tmpl, d_ok, p, &o, s))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	unsigned long long c[8];
	HIP_TRY(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost));
	if(c[1]) HIP_TRY(hipMemset(ws->counters + 1, 0, sizeof(unsigned long long)));
	if(ops_needed) *ops_needed = (int64_t) c[0];
	if(c[1] == 2 || (int64_t) c[0] > out->ops_cap) { kmahip_set_error("ops_cap %lld too small, need %llu", (long long) out->ops_cap, c[0]); return KMAHIP_EOVERFLOW; }
	if(c[1]) { kmahip_set_error("trace stage: a read needs more scratch than the workspace holds (status %llu)", c[1]); return KMAHIP_EDEVICE; }
	HIP_TRY(hipMemcpy(out->ops_off, o.ops_off, (size_t) n * 8, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(out->stats, o.stats, (size_t) n * 40, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(out->n_ops, o.n_ops, (size_t) n * 4, hipMemcpyDeviceToHost));
	if(c[0]) HIP_TRY(hipMemcpy(out->ops, o.ops, (size_t) c[0] * 4, hipMemcpyDeviceToHost));
	return KMAHIP_OK;
}

extern "C" int kmahip_align_trace_mt1_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, int32_t tmpl, int one2one, const kmahip_params *p,
                                          kmahip_traces *out, int32_t *rc_out, void *stream) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_longtrace(db, ws, reads, nullptr, tmpl, nullptr, nullptr, one2one, p, out, rc_out, (hipStream_t) stream);
}

extern "C" int kmahip_align_trace_mt1(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads, int32_t tmpl, int one2one, const kmahip_params *p,
                                      kmahip_traces *out, int32_t *rc_out, int64_t *ops_needed) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int64_t n = reads->n_reads;
	if(n < 0 || reads->seq_words < 0 || reads->N_total < 0 || out->ops_cap < 0) { kmahip_set_error("negative size"); return KMAHIP_EINVAL; }
	if(ops_needed) *ops_needed = 0;
	if(n == 0) return KMAHIP_OK;
	int rc;
	if((rc = stage_reserve(ws, 0, (size_t) (reads->seq_words + 2) * 8)) || (rc = stage_reserve(ws, 1, (size_t) (n + 1) * 8)) ||
	   (rc = stage_reserve(ws, 2, (size_t) n * 4)) || (rc = stage_reserve(ws, 3, (size_t) reads->N_total * 4)) ||
	   (rc = stage_reserve(ws, 4, (size_t) (n + 1) * 8)) || (rc = stage_reserve(ws, 5, (size_t) n * 4 + 8)) ||
	   (rc = stage_reserve(ws, 6, (size_t) n * (40 + 8 + 4) + 8)) || (rc = stage_reserve(ws, 7, (size_t) (out->ops_cap + 1) * 4))) return rc;
	hipStream_t s = 0;
	HIP_TRY(hipMemsetAsync((char *) ws->stage[0] + (size_t) reads->seq_words * 8, 0, 16, s));
	if(reads->seq_words) HIP_TRY(hipMemcpyAsync(ws->stage[0], reads->seq, (size_t) reads->seq_words * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[1], reads->seq_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[2], reads->len, (size_t) n * 4, hipMemcpyHostToDevice, s));
	if(reads->N_total) HIP_TRY(hipMemcpyAsync(ws->stage[3], reads->N, (size_t) reads->N_total * 4, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemcpyAsync(ws->stage[4], reads->N_off, (size_t) (n + 1) * 8, hipMemcpyHostToDevice, s));
	kmahip_reads d = *reads;
	d.q_start = nullptr; d.q_end = nullptr;       // (host pointers, if any: this entry point maps whole reads)
	d.seq = (const uint64_t *) ws->stage[0]; d.seq_off = (const int64_t *) ws->stage[1]; d.len = (const int32_t *) ws->stage[2];
	d.N = (const int32_t *) ws->stage[3]; d.N_off = (const int64_t *) ws->stage[4];
	int32_t *d_rc = (int32_t *) ws->stage[5];
	kmahip_traces o;
	o.ops_off = (int64_t *) ws->stage[6]; o.stats = (int32_t *) (o.ops_off + n); o.n_ops = o.stats + 10 * n;
	o.ops = (uint32_t *) ws->stage[7]; o.ops_cap = out->ops_cap;
	if((rc = kmahip_launch_longtrace(db, ws, &d, nullptr, tmpl, nullptr, nullptr, one2one, p, &o, d_rc, s))) return rc;
	HIP_TRY(hipStreamSynchronize(s));
	unsigned long long c[2];
	HIP_TRY(hipMemcpy(c, ws->counters, sizeof c, hipMemcpyDeviceToHost));
	if(c[1]) HIP_TRY(hipMemset(ws->counters + 1, 0, sizeof(unsigned long long)));
	if(ops_needed) *ops_needed = (int64_t) c[0];
	if(c[1] == 2 || (int64_t) c[0] > out->ops_cap) { kmahip_set_error("ops_cap %lld too small, need %llu", (long long) out->ops_cap, c[0]); return KMAHIP_EOVERFLOW; }
	HIP_TRY(hipMemcpy(out->ops_off, o.ops_off, (size_t) n * 8, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(out->stats, o.stats, (size_t) n * 40, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(out->n_ops, o.n_ops, (size_t) n * 4, hipMemcpyDeviceToHost));
	if(rc_out) HIP_TRY(hipMemcpy(rc_out, d_rc, (size_t) n * 4, hipMemcpyDeviceToHost));
	if(c[0]) HIP_TRY(hipMemcpy(out->ops, o.ops, (size_t) c[0] * 4, hipMemcpyDeviceToHost));
	return KMAHIP_OK;
}

extern "C" int kmahip_trace_get_stats(kmahip_ws *ws, kmahip_trace_stats *st) {
	if(!ws || !st) return KMAHIP_EINVAL;
	st->problems = ws->lt_stats[0]; st->dp_cells = ws->lt_stats[1]; st->mems = ws->lt_stats[2]; st->reads = ws->lt_stats[3];
	return KMAHIP_OK;
}

// kmahip.h: reads of the default mode whose result the reference takes from memory it never cleared
extern "C" int kmahip_chain_unpinned_reads(const int32_t *len, const int32_t *N, const int64_t *N_off, int64_t n_reads, int k,
                                           int32_t longest_before, int32_t *longest_after, int64_t *count) {
	if(!len || !N_off || n_reads < 0 || k < 1 || !count) { kmahip_set_error("bad arguments"); return KMAHIP_EINVAL; }
	int32_t longest = longest_before;
	int64_t c = 0;
	for(int64_t r = 0; r < n_reads; ++r) {
		if(N_off[r + 1] > N_off[r] && N && N[N_off[r]] < k - 1 && longest > len[r]) ++c;      // (a read's N positions ascend)
		if(len[r] > longest) longest = len[r];
	}
	if(longest_after) *longest_after = longest;
	*count = c;
	return KMAHIP_OK;
}
