// api.hip -- extern "C" entry points of libkmahip.so (see include/kmahip.h).
#include "kmahip_internal.h"
#include <cstring>

extern "C" int kmahip_ws_create(kmahip_db *db, kmahip_ws **out) {
	if(!db || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	kmahip_ws *ws = new kmahip_ws();
	memset(ws, 0, sizeof *ws);
	ws->db = db;
	*out = ws;
	return KMAHIP_OK;
}

extern "C" void kmahip_ws_destroy(kmahip_ws *ws) {
	if(!ws) return;
	(void) hipFree(ws->item_score); (void) hipFree(ws->item_n); (void) hipFree(ws->item_off);
	(void) hipFree(ws->pool); (void) hipFree(ws->counters); (void) hipFree(ws->overflow_items);
	(void) hipFree(ws->dense); (void) hipFree(ws->blk_sums);
	for(int i = 0; i < 8; ++i) (void) hipFree(ws->stage[i]);
	if(ws->events) {
		for(auto &e : *ws->events) { (void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second); }
		delete ws->events;
	}
	delete ws;
}

extern "C" int kmahip_scan_se_dev(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                                  const kmahip_params *p, kmahip_cands *out, void *stream) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_launch_scan_se(db, ws, reads, p, out, (hipStream_t) stream);
}

extern "C" int kmahip_ws_status(kmahip_ws *ws, void *stream) {
	if(!ws || !ws->counters) return KMAHIP_EINVAL;
	unsigned long long c[8];
	HIP_TRY(hipMemcpyAsync(c, ws->counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t) stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
	if(c[1]) { kmahip_set_error("output capacity too small (status %llu)", c[1]); return KMAHIP_EOVERFLOW; }
	return KMAHIP_OK;
}

extern "C" int kmahip_scan_set_stats(kmahip_ws *ws, int on) {
	if(!ws) return KMAHIP_EINVAL;
	ws->stats_on = on;
	return KMAHIP_OK;
}

extern "C" int kmahip_scan_get_stats(kmahip_ws *ws, kmahip_scan_stats *st, void *stream) {
	if(!ws || !st || !ws->counters) return KMAHIP_EINVAL;
	unsigned long long c[8];
	HIP_TRY(hipMemcpyAsync(c, ws->counters, sizeof c, hipMemcpyDeviceToHost, (hipStream_t) stream));
	HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
	st->probes = c[3]; st->value_elems = c[4]; st->active_strands = c[5];
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

extern "C" int kmahip_scan_se(kmahip_db *db, kmahip_ws *ws, const kmahip_reads *reads,
                              const kmahip_params *p, kmahip_cands *out) {
	if(!db || !ws || !reads || !p || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
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
	if(c[1] == 1) {
		// internal candidate pool too small: grow and let the caller retry
		ws->cap_reads = 0;
		kmahip_set_error("internal candidate pool exhausted");
		return KMAHIP_EOVERFLOW;
	}
	const int64_t total = out->T_off[n];
	if(total > out->T_cap) { kmahip_set_error("T_cap %lld too small, need %lld", (long long) out->T_cap, (long long) total); return KMAHIP_EOVERFLOW; }
	if(total) HIP_TRY(hipMemcpy(out->T, o.T, (size_t) total * 4, hipMemcpyDeviceToHost));
	return KMAHIP_OK;
}

extern "C" int kmahip_ws_set_timing(kmahip_ws *ws, int on) {
	if(!ws) return KMAHIP_EINVAL;
	ws->timing_on = on;
	return KMAHIP_OK;
}

extern "C" int kmahip_ws_get_timing(kmahip_ws *ws, double *total_ms, int64_t *launches) {
	if(!ws || !total_ms || !launches) return KMAHIP_EINVAL;
	*total_ms = 0.0; *launches = 0;
	if(!ws->events) return KMAHIP_OK;
	for(auto &e : *ws->events) {
		float ms = 0.f;
		HIP_TRY(hipEventSynchronize(e.second));
		HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
		*total_ms += ms; *launches += 1;
		(void) hipEventDestroy(e.first); (void) hipEventDestroy(e.second);
	}
	ws->events->clear();
	return KMAHIP_OK;
}
