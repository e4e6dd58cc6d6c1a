// fragout.hip -- the `.frag.gz` rows of stage 3c (egress half), host code.
// Restated from updateFrags (assembly.c:49-83) and the loop of assemble_KMA that calls it (assembly.c:1890-2000): one row
// per read that passed the stage-3c filter -- the read as it was aligned (reverse complemented when it was filed on the
// minus strand), the number of equally good templates, score, start, end, template name, read header.
#include "kmahip_internal.h"
#include <zlib.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace {

int load_names(kmahip_db *db) {
	if(!db->h_names.empty()) return KMAHIP_OK;
	FILE *f = fopen((db->prefix + ".name").c_str(), "rb");
	if(!f) { kmahip_set_error("cannot open %s.name", db->prefix.c_str()); return KMAHIP_EIO; }
	std::string line;
	int c;
	while((c = fgetc(f)) != EOF) {
		if(c == '\n') { db->h_names.push_back(line); line.clear(); }
		else line.push_back((char) c);
	}
	if(!line.empty()) db->h_names.push_back(line);
	fclose(f);
	return KMAHIP_OK;
}

}  // namespace

extern "C" int kmahip_frag_write(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                                 const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, const char *read_names,
                                 const int64_t *read_name_off, int64_t *rows) {
	return kmahip_frag_write2(path, db, reads, rc, tmpl, n_hits, trace_stats, max_frag, 0, read_names, read_name_off, rows);
}

extern "C" int kmahip_frag_write2(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                                  const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, int order, const char *read_names,
                                  const int64_t *read_name_off, int64_t *rows) {
	if(!path || !db || !reads || !rc || !tmpl || !n_hits || !trace_stats || !read_names || !read_name_off) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	int e = load_names(db);
	if(e) return e;
	if(max_frag <= 0) max_frag = 1000000;
	const int64_t n = reads->n_reads;
	// the order assemble_KMA meets the fragments in: templates ascending; inside a template the chunks of max_frag filed
	// fragments in stream order, each chunk back to front (conclave.c:164-166, 194)
	struct Key { int32_t t; int64_t chunk, rank, read; };
	std::vector<Key> keys;
	int64_t rank = 0;
	for(int64_t i = 0; i < n; ++i) {
		if(tmpl[i] == 0) continue;
		const int64_t r = rank++;
		if(trace_stats[10 * i + 3] == 0) continue;          // dropped by the stage-3c filter: no row
		keys.push_back(Key{abs(tmpl[i]), order ? 0 : r / max_frag, order ? -r : r, i});      // (order 1: the single thread of `-Mt1` writes them as they come)
	}
	std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
		if(a.t != b.t) return a.t < b.t;
		if(a.chunk != b.chunk) return a.chunk < b.chunk;
		return a.rank > b.rank;
	});
	const size_t plen = strlen(path);
	const bool gz = plen > 3 && !strcmp(path + plen - 3, ".gz");
	gzFile g = gzopen(path, gz ? "wb1" : "wbT");            // level 1 like the reference's deflateInit2 (filebuff.c:189); T = plain
	if(!g) { kmahip_set_error("cannot create %s", path); return KMAHIP_EIO; }
	gzbuffer(g, 1 << 20);
	static const char bases[] = "ACGTN";
	std::string row;
	for(const Key &k : keys) {
		const int64_t i = k.read;
		const int L = reads->len[i];
		const uint64_t *w = reads->seq + reads->seq_off[i];
		const int32_t *N = reads->N + reads->N_off[i];
		const int nN = (int) (reads->N_off[i + 1] - reads->N_off[i]);
		row.assign((size_t) L, 'A');
		for(int p = 0; p < L; ++p) row[(size_t) p] = bases[(w[p >> 5] >> (62 - ((p & 31) << 1))) & 3];
		for(int x = 0; x < nN; ++x) row[(size_t) N[x]] = 'N';
		const bool flip = ((rc[i] & 1) != 0) != (tmpl[i] < 0);
		if(flip) {
			std::reverse(row.begin(), row.end());
			for(char &c : row) c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
		}
		if((size_t) k.t > db->h_names.size()) { gzclose(g); kmahip_set_error("template %d has no name in %s.name", k.t, db->prefix.c_str()); return KMAHIP_EFORMAT; }
		char num[96];
		snprintf(num, sizeof num, "\t%d\t%d\t%d\t%d\t", n_hits[i], trace_stats[10 * i], trace_stats[10 * i + 1], trace_stats[10 * i + 2]);
		row += num;
		row += db->h_names[(size_t) k.t - 1];
		row += '\t';
		row.append(read_names + read_name_off[i]);              // NUL-terminated
		row += '\n';
		if(gzwrite(g, row.data(), (unsigned) row.size()) != (int) row.size()) { gzclose(g); kmahip_set_error("write to %s failed", path); return KMAHIP_EIO; }
	}
	if(gzclose(g) != Z_OK) { kmahip_set_error("closing %s failed", path); return KMAHIP_EIO; }
	if(rows) *rows = (int64_t) keys.size();
	return KMAHIP_OK;
}
