// fragout.hip -- the `.frag.gz` rows of stage 3c (egress half), host code.
// Restated from updateFrags (assembly.c:49-83) and the loop of assemble_KMA that calls it (assembly.c:1890-2000): one row
// per read that passed the stage-3c filter -- the read as it was aligned (reverse complemented when it was filed on the
// minus strand), the number of equally good templates, score, start, end, template name, read header.
#include "kmahip_internal.h"
#include "fastgz.h"
#include <zlib.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
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

// Rows [0, n_rows) written to `path` in order. The rows are cut into blocks; a few threads format blocks side by side and, for
// a .gz, compress each one as a gzip member of its own (fastgz.h: entropy coding only, where the reference runs zlib at level
// 1, filebuff.c:189); the caller's thread writes the finished blocks in order. A file of concatenated members is a gzip file (RFC 1952 2.2): zcat, gzread and
// the reference's own reader give back the same bytes as one stream would.
struct RowBlock { std::string data; bool ready = false; };

template <class Fmt>
int write_rows(const char *path, size_t n_rows, size_t block_rows, Fmt fmt) {
	const size_t plen = strlen(path);
	const bool gz = plen > 3 && !strcmp(path + plen - 3, ".gz");
	FILE *f = fopen(path, "wb");
	if(!f) { kmahip_set_error("cannot create %s", path); return KMAHIP_EIO; }
	const size_t BLOCK = std::max<size_t>(1, std::min<size_t>(16384, block_rows));      // rows per block: some megabytes of text
	const size_t n_blocks = (n_rows + BLOCK - 1) / BLOCK;
	const char *e = getenv("KMAHIP_IO_THREADS");
	const int hw = (int) std::thread::hardware_concurrency();
	int nt = e ? atoi(e) : std::min(16, hw > 0 ? hw : 1);
	nt = (int) std::max<size_t>(1, std::min<size_t>((size_t) std::max(nt, 1), n_blocks));
	const size_t WINDOW = (size_t) nt * 4;                // blocks in flight: bounds the memory held
	std::vector<RowBlock> ring(WINDOW);
	std::mutex mu;
	std::condition_variable cv;
	size_t written = 0;                                  // blocks [0, written) are on disk and their ring slots free
	std::atomic<size_t> next{0};
	std::atomic<int> failed{0};
	std::atomic<long long> us_fmt{0}, us_zip{0};
	const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto us = [](auto a, auto b) { return (long long) std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
	const auto t_begin = now();
	long long us_write = 0;
	auto worker = [&]() {
		std::string raw;
		std::vector<uint8_t> scratch;
		for(;;) {
			const size_t b = next.fetch_add(1);
			if(b >= n_blocks || failed.load()) return;
			{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return b < written + WINDOW || failed.load(); }); }
			if(failed.load()) return;
			raw.clear();
			const auto t0 = now();
			const size_t r1 = std::min(n_rows, (b + 1) * BLOCK);
			for(size_t r = b * BLOCK; r < r1; ++r) fmt(r, raw);
			const auto t1 = now();
			RowBlock &B = ring[b % WINDOW];
			if(gz) {
				B.data.clear();
				fastgz::gzip_member((const uint8_t *) raw.data(), raw.size(), B.data, scratch);
			} else B.data.swap(raw);
			us_fmt += us(t0, t1); us_zip += us(t1, now());
			{ std::lock_guard<std::mutex> lk(mu); B.ready = true; }
			cv.notify_all();
		}
	};
	std::vector<std::thread> pool;
	for(int t = 0; t < nt; ++t) pool.emplace_back(worker);
	int rc = KMAHIP_OK;
	for(size_t b = 0; b < n_blocks; ++b) {
		RowBlock &B = ring[b % WINDOW];
		{ std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return B.ready || failed.load(); }); }
		if(failed.load()) { rc = KMAHIP_EIO; break; }
		const auto t0 = now();
		if(fwrite(B.data.data(), 1, B.data.size(), f) != B.data.size()) { failed.store(1); rc = KMAHIP_EIO; }
		us_write += us(t0, now());
		{ std::lock_guard<std::mutex> lk(mu); B.ready = false; written = b + 1; }
		cv.notify_all();
		if(rc) break;
	}
	cv.notify_all();
	for(std::thread &th : pool) th.join();
	if(n_blocks == 0 && gz) {                             // an empty .gz is still a gzip file
		gzFile g = gzdopen(dup(fileno(f)), "wb1");
		if(!g || gzclose(g) != Z_OK) rc = KMAHIP_EIO;
	}
	if(fclose(f) != 0) rc = KMAHIP_EIO;
	if(dbg) fprintf(stderr, "[kmahip] write_rows: %zu rows, %d threads: wall %.1f ms; formatting %.1f ms and compressing %.1f ms summed over the threads, writing %.1f ms\n",
	                n_rows, nt, us(t_begin, now()) / 1e3, us_fmt.load() / 1e3, us_zip.load() / 1e3, us_write / 1e3);
	if(rc) kmahip_set_error("write to %s failed", path);
	return rc;
}

}  // namespace

// the template names of <prefix>.name, one per template in template order (read once)
int kmahip_db_load_names(kmahip_db *db) { return load_names(db); }

// ---- text that is formatted elsewhere (the session's rows come off the device as text), handed over block by block IN ORDER:
// a few threads compress the blocks, each a gzip member of its own (fastgz.h), a writer thread puts them into the file in order.
// The text is not copied: `done` of a block is counted down when its bytes are no longer needed.
struct kmahip_gzstream {
	struct Blk { const char *text; size_t bytes; std::atomic<int> *done; std::string out; int state; };      // state 0 queued, 1 taken, 2 ready
	FILE *f = nullptr;
	bool gz = false;
	std::deque<Blk> q;                  // blocks [base, base + q.size())
	size_t base = 0, next_work = 0;     // next_work: absolute index of the first block no worker has taken
	bool closing = false;
	int rc = KMAHIP_OK;
	std::mutex mu;
	std::condition_variable cv;
	std::vector<std::thread> workers;
	std::thread writer;
	size_t blocks = 0;
};

static void gzstream_work(kmahip_gzstream *g) {
	std::vector<uint8_t> scratch;
	std::string out;
	for(;;) {
		kmahip_gzstream::Blk *b = nullptr;
		size_t idx = 0;
		{
			std::unique_lock<std::mutex> lk(g->mu);
			g->cv.wait(lk, [&] { return g->next_work < g->base + g->q.size() || g->closing; });
			if(g->next_work >= g->base + g->q.size()) return;
			idx = g->next_work++;
			b = &g->q[idx - g->base];
			b->state = 1;
		}
		out.clear();
		if(g->gz) fastgz::gzip_member((const uint8_t *) b->text, b->bytes, out, scratch);
		else out.assign(b->text, b->bytes);
		{
			std::lock_guard<std::mutex> lk(g->mu);
			kmahip_gzstream::Blk &bb = g->q[idx - g->base];          // (the deque may have grown: take the block again)
			bb.out.swap(out);
			bb.state = 2;
			if(bb.done) bb.done->fetch_sub(1);
		}
		g->cv.notify_all();
	}
}

static void gzstream_write(kmahip_gzstream *g) {
	for(;;) {
		std::string out;
		{
			std::unique_lock<std::mutex> lk(g->mu);
			g->cv.wait(lk, [&] { return (!g->q.empty() && g->q.front().state == 2) || (g->closing && g->q.empty()); });
			if(g->q.empty()) return;
			out.swap(g->q.front().out);
			g->q.pop_front();
			++g->base;
		}
		g->cv.notify_all();
		if(fwrite(out.data(), 1, out.size(), g->f) != out.size()) { std::lock_guard<std::mutex> lk(g->mu); g->rc = KMAHIP_EIO; }
	}
}

kmahip_gzstream *kmahip_gzstream_open(const char *path) {
	FILE *f = fopen(path, "wb");
	if(!f) { kmahip_set_error("cannot create %s", path); return nullptr; }
	kmahip_gzstream *g = new kmahip_gzstream();
	g->f = f;
	const size_t plen = strlen(path);
	g->gz = plen > 3 && !strcmp(path + plen - 3, ".gz");
	const char *e = getenv("KMAHIP_IO_THREADS");
	const int hw = (int) std::thread::hardware_concurrency();
	int nt = e ? atoi(e) : std::min(16, hw > 0 ? hw : 1);
	nt = std::max(nt, 1);
	for(int t = 0; t < nt; ++t) g->workers.emplace_back(gzstream_work, g);
	g->writer = std::thread(gzstream_write, g);
	return g;
}

// the next block of the file; `done` (may be NULL) is counted down by one when the block's text has been read for the last time
void kmahip_gzstream_submit(kmahip_gzstream *g, const char *text, size_t bytes, std::atomic<int> *done) {
	{
		std::lock_guard<std::mutex> lk(g->mu);
		g->q.push_back({text, bytes, done, std::string(), 0});
		++g->blocks;
	}
	g->cv.notify_all();
}

int kmahip_gzstream_close(kmahip_gzstream *g) {
	{ std::lock_guard<std::mutex> lk(g->mu); g->closing = true; }
	g->cv.notify_all();
	for(std::thread &t : g->workers) t.join();
	g->writer.join();
	int rc = g->rc;
	if(g->blocks == 0 && g->gz) {                             // an empty .gz is still a gzip file
		gzFile z = gzdopen(dup(fileno(g->f)), "wb1");
		if(!z || gzclose(z) != Z_OK) rc = KMAHIP_EIO;
	}
	if(fclose(g->f) != 0) rc = KMAHIP_EIO;
	if(rc) kmahip_set_error("write to the fragment file failed");
	delete g;
	return rc;
}

int kmahip_frag_write_src(const char *path, kmahip_db *db, const kmahip_reads *reads, int64_t n, const int64_t *src, const int32_t *rc,
                          const int32_t *tmpl, const int32_t *n_hits, const int32_t *trace_stats, int stats_stride, int64_t max_frag, int order,
                          const int64_t *frag_rank, const char *read_names, const int64_t *read_name_off, int64_t *rows);

extern "C" int kmahip_frag_write(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                                 const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, const char *read_names,
                                 const int64_t *read_name_off, int64_t *rows) {
	return kmahip_frag_write2(path, db, reads, rc, tmpl, n_hits, trace_stats, max_frag, 0, read_names, read_name_off, rows);
}

extern "C" int kmahip_frag_write2(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                                  const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, int order, const char *read_names,
                                  const int64_t *read_name_off, int64_t *rows) {
	return kmahip_frag_write3(path, db, reads, rc, tmpl, n_hits, trace_stats, max_frag, order, nullptr, read_names, read_name_off, rows);
}

extern "C" int kmahip_frag_write3(const char *path, kmahip_db *db, const kmahip_reads *reads, const int32_t *rc, const int32_t *tmpl,
                                  const int32_t *n_hits, const int32_t *trace_stats, int64_t max_frag, int order, const int64_t *frag_rank,
                                  const char *read_names, const int64_t *read_name_off, int64_t *rows) {
	if(!reads) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	return kmahip_frag_write_src(path, db, reads, reads->n_reads, nullptr, rc, tmpl, n_hits, trace_stats, 10, max_frag, order, frag_rank, read_names, read_name_off, rows);
}

// the writer proper. src == NULL: fragment i is read i of `reads`; else fragment i (of n) is read src[i] of `reads` / `read_names`
// (the paired run files its fragments in record order without copying the reads). rc, tmpl, n_hits, trace_stats, frag_rank: per fragment.
int kmahip_frag_write_src(const char *path, kmahip_db *db, const kmahip_reads *reads, int64_t n, const int64_t *src, const int32_t *rc,
                          const int32_t *tmpl, const int32_t *n_hits, const int32_t *trace_stats, int stats_stride, int64_t max_frag, int order,
                          const int64_t *frag_rank, const char *read_names, const int64_t *read_name_off, int64_t *rows) {
	const int64_t S = stats_stride;         // ints per fragment in trace_stats (score, start, end, kept first)
	if(!path || !db || !reads || n < 0 || (n > 0 && (!rc || !tmpl || !n_hits || !trace_stats || !read_names || !read_name_off))) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	int e = load_names(db);
	if(e) return e;
	if(max_frag <= 0) max_frag = 1000000;
	auto rd = [src](int64_t i) { return src ? src[i] : i; };
	// the order assemble_KMA meets the fragments in: templates ascending; inside a template the chunks of max_frag filed
	// fragments in stream order, each chunk back to front (conclave.c:164-166, 194). A counting sort over the templates keeps
	// the stream order inside each one; the chunks are then turned round in place. (order 1: the single thread of `-Mt1`
	// writes the rows as they come.)
	const auto t_order = std::chrono::steady_clock::now();
	const size_t n_t = db->h_names.size();
	// (the fragments are cut into one contiguous range per thread: counts per range and template, a prefix over (template, range),
	// then every range drops its rows where they belong -- the stream order inside a template is kept)
	int nt;
	const int64_t grain = getenv("KMAHIP_ROW_GRAIN") ? std::max(1, atoi(getenv("KMAHIP_ROW_GRAIN"))) : 65536;      // fragments per thread at least (the tests lower it)
	{
		const char *e_ = getenv("KMAHIP_IO_THREADS");
		const int hw = (int) std::thread::hardware_concurrency();
		nt = e_ ? atoi(e_) : std::min(16, hw > 0 ? hw : 1);
		nt = (int) std::max<int64_t>(1, std::min<int64_t>(std::max(nt, 1), n / grain));
	}
	auto lo = [&](int w) { return n * w / nt; };
	auto in_threads = [&](auto fn) {
		std::vector<std::thread> pool;
		for(int w = 1; w < nt; ++w) pool.emplace_back(fn, w);
		fn(0);
		for(std::thread &th : pool) th.join();
	};
	std::vector<std::vector<int64_t>> cnt((size_t) nt, std::vector<int64_t>(n_t + 2, 0));
	std::vector<int64_t> filed((size_t) nt + 1, 0);
	std::atomic<size_t> bad_t{0};
	in_threads([&](int w) {
		std::vector<int64_t> &c = cnt[(size_t) w];
		int64_t f = 0;
		for(int64_t i = lo(w); i < lo(w + 1); ++i) {
			if(tmpl[i] == 0) continue;
			const size_t t = (size_t) abs(tmpl[i]);
			if(t > n_t) { bad_t.store(t); return; }
			++f;
			if(trace_stats[S * i + 3] != 0) ++c[t];
		}
		filed[(size_t) w + 1] = f;
	});
	if(bad_t.load()) { kmahip_set_error("template %zu has no name in %s.name", bad_t.load(), db->prefix.c_str()); return KMAHIP_EFORMAT; }
	std::vector<int64_t> t_rows(n_t + 2, 0);            // t_rows[t] = first row of template t
	{
		int64_t at = 0;
		for(size_t t = 0; t <= n_t; ++t) {
			t_rows[t] = at;
			for(int w = 0; w < nt; ++w) { const int64_t c = cnt[(size_t) w][t]; cnt[(size_t) w][t] = at; at += c; }    // now: where range w starts filling t
		}
		t_rows[n_t + 1] = at;
		for(int w = 0; w < nt; ++w) filed[(size_t) w + 1] += filed[(size_t) w];
	}
	const size_t n_rows = (size_t) t_rows[n_t + 1];
	std::vector<int64_t> row_read(n_rows), row_rank(order ? 0 : n_rows);
	in_threads([&](int w) {
		std::vector<int64_t> &fill = cnt[(size_t) w];
		int64_t rank = filed[(size_t) w];                  // position among the filed reads of the whole run
		for(int64_t i = lo(w); i < lo(w + 1); ++i) {
			if(tmpl[i] == 0) continue;
			const size_t t = (size_t) abs(tmpl[i]);
			const int64_t r = frag_rank ? frag_rank[i] : rank++;
			if(trace_stats[S * i + 3] == 0) continue;          // dropped by the stage-3c filter: no row
			const size_t at = (size_t) fill[t]++;
			row_read[at] = i;
			if(!order) row_rank[at] = r;
		}
	});
	if(!order) {
		std::atomic<size_t> next_t{1};
		const int nt0 = nt;
		nt = (int) std::max<size_t>(1, std::min<size_t>((size_t) nt0, n_rows / (size_t) grain));
		in_threads([&](int) {
			for(;;) {
				const size_t t0 = next_t.fetch_add(64);
				if(t0 > n_t) return;
				for(size_t t = t0; t <= std::min(n_t, t0 + 63); ++t) {
					size_t a = (size_t) t_rows[t];
					const size_t b = (size_t) t_rows[t + 1];
					while(a < b) {
						const int64_t chunk = row_rank[a] / max_frag;
						size_t c = a + 1;
						while(c < b && row_rank[c] / max_frag == chunk) ++c;
						std::reverse(row_read.begin() + (ptrdiff_t) a, row_read.begin() + (ptrdiff_t) c);
						a = c;
					}
				}
			}
		});
	}
	if(getenv("KMAHIP_DEBUG_TIMING")) fprintf(stderr, "[kmahip] frag_write: row order %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_order).count());
	// The rows come in template order, the reads lie in stream order: every row gathers from a dozen cache lines nobody has
	// touched lately. The lines of the rows to come are asked for ahead of time: the per-read entries 16 rows ahead, what they
	// point to (bases, header) 8 rows ahead.
	auto ahead = [&](size_t r) {
		if(r + 16 < n_rows) {
			const int64_t i = row_read[r + 16], j = rd(i);
			__builtin_prefetch(reads->seq_off + j); __builtin_prefetch(reads->len + j); __builtin_prefetch(reads->N_off + j);
			__builtin_prefetch(rc + i); __builtin_prefetch(tmpl + i); __builtin_prefetch(n_hits + i);
			__builtin_prefetch(trace_stats + S * i); __builtin_prefetch(read_name_off + j);
		}
		if(r + 8 < n_rows) {
			const int64_t i = row_read[r + 8], j = rd(i);
			const uint64_t *w = reads->seq + reads->seq_off[j];
			__builtin_prefetch(w); __builtin_prefetch(w + 8);
			__builtin_prefetch(read_names + read_name_off[j]);
			__builtin_prefetch(db->h_names[(size_t) abs(tmpl[i]) - 1].data());
		}
	};
	struct Four {
		uint32_t t[256];
		char c[256];
		Four() {
			for(int v = 0; v < 256; ++v) {
				char b[4];
				for(int j = 0; j < 4; ++j) b[j] = "ACGT"[(v >> (6 - 2 * j)) & 3];
				memcpy(&t[v], b, 4);
				c[v] = (char) v;
			}
			c[(int) 'A'] = 'T'; c[(int) 'C'] = 'G'; c[(int) 'G'] = 'C'; c[(int) 'T'] = 'A';
		}
	};
	static const Four four;
	auto put_int = [](char *o, int v) {          // "\t<v>", returns the end
		*o++ = '\t';
		unsigned u = v < 0 ? 0u - (unsigned) v : (unsigned) v;
		if(v < 0) *o++ = '-';
		char d[12];
		int k = 0;
		do { d[k++] = (char) ('0' + u % 10); u /= 10; } while(u);
		while(k) *o++ = d[--k];
		return o;
	};
	auto fmt = [&](size_t r, std::string &out) {
		ahead(r);
		const int64_t i = row_read[r], j = rd(i);
		const int L = reads->len[j];
		const uint64_t *w = reads->seq + reads->seq_off[j];
		const int32_t *N = reads->N + reads->N_off[j];
		const int nN = (int) (reads->N_off[j + 1] - reads->N_off[j]);
		const bool flip = ((rc[i] & 1) != 0) != (tmpl[i] < 0);
		const std::string &tname = db->h_names[(size_t) abs(tmpl[i]) - 1];
		const char *rname = read_names + read_name_off[j];            // NUL-terminated
		const size_t rlen = strlen(rname);
		const size_t at = out.size();
		out.resize(at + (size_t) (((L + 31) >> 5) << 5) + 4 * 12 + 2 + tname.size() + rlen + 1);
		char *o = &out[at];
		{	// four bases per table look-up (the room behind the bases takes the overshoot of the last one)
			const int words = (L + 31) >> 5;
			char *d = o;
			for(int x = 0; x < words; ++x) {
				const uint64_t v = w[x];
				for(int b = 56; b >= 0; b -= 8, d += 4) memcpy(d, &four.t[(v >> b) & 0xff], 4);
			}
			if(flip) {
				for(int a = 0, b = L - 1; a < b; ++a, --b) { const char ca = o[a], cb = o[b]; o[a] = four.c[(unsigned char) cb]; o[b] = four.c[(unsigned char) ca]; }
				if(L & 1) o[L >> 1] = four.c[(unsigned char) o[L >> 1]];
			}
		}
		for(int x = 0; x < nN; ++x) o[flip ? L - 1 - N[x] : N[x]] = 'N';
		o += L;
		o = put_int(o, n_hits[i]); o = put_int(o, trace_stats[S * i]); o = put_int(o, trace_stats[S * i + 1]); o = put_int(o, trace_stats[S * i + 2]);
		*o++ = '\t';
		memcpy(o, tname.data(), tname.size()); o += tname.size();
		*o++ = '\t';
		memcpy(o, rname, rlen); o += rlen;
		*o++ = '\n';
		out.resize((size_t) (o - out.data()));
	};
	// (16 384 rows of short reads are 5 MB of text; rows of long reads are cut into blocks of about that size too, so that a few
	// thousand 10 kb rows still keep every thread busy)
	size_t row_bytes = 64;
	if(n_rows) {
		const size_t step = std::max<size_t>(1, n_rows / 1024);
		size_t sum = 0, cnt = 0;
		for(size_t r = 0; r < n_rows; r += step, ++cnt) sum += (size_t) reads->len[rd(row_read[r])];
		row_bytes += sum / cnt;
	}
	if((e = write_rows(path, n_rows, (5u << 20) / row_bytes, fmt))) return e;
	if(rows) *rows = (int64_t) n_rows;
	return KMAHIP_OK;
}

// one gzip member as the writers above make them (for the tests of fastgz.h; KMAHIP_EINVAL when dst is too small)
extern "C" int kmahip_gzip_member(const void *src, int64_t n, void *dst, int64_t cap, int64_t *out_bytes) {
	if((!src && n) || !dst || !out_bytes || n < 0) { kmahip_set_error("bad argument"); return KMAHIP_EINVAL; }
	std::string o;
	fastgz::gzip_member((const uint8_t *) src, (size_t) n, o);
	if((int64_t) o.size() > cap) { kmahip_set_error("destination too small: %zu bytes needed", o.size()); return KMAHIP_EINVAL; }
	memcpy(dst, o.data(), o.size());
	*out_bytes = (int64_t) o.size();
	return KMAHIP_OK;
}
