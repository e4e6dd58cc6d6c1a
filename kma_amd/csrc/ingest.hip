// ingest.hip -- stage 1 of KMA on the host: FASTQ / FASTA records -> trimmed, 2-bit packed read batches.
// Behaviour restated from run_input / run_input_PE (runinput.c:370-606), phredStat / fsastat (:127-368),
// FileBuffgetFq / FileBuffgetFsa (seqparse.c:241-403, 66-159), getPhredFileBuff (:551-589), the to2Bit table
// (kma.c:1440-1480) and compDNA (compdna.c:99-127). Host-only translation unit; nothing here touches the device.
#include "kmahip_internal.h"
#include <zlib.h>
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t FIRST_CHUNK = 1048576;     // the reference guesses the phred scale from its first file buffer (filebuff.h:36)

struct Trans {
	uint8_t t[256];
	Trans() {
		memset(t, 8, sizeof t);
		t[(int) '\n'] = 16;
		const char *codes[5] = {"ARMDrmd", "CYBcyb", "GSKVgskv", "TWHUtwh", "NXnx"};
		for(int c = 0; c < 5; ++c) for(const char *p = codes[c]; *p; ++p) t[(int) (unsigned char) *p] = (uint8_t) c;
		t[(int) 'a'] = 0; t[(int) 'u'] = 8;          // lower-case u is not in the reference's table
	}
};
const Trans g_trans;

// 10^(-q/10) for q = 0 .. 255. The reference's table (kma.c:219) holds 32-decimal literals, i.e. pow(10, -0.1 q) printed
// with "%.32f" and read back: exact doubles down to 1e-16, truncated decimals below. Generated the same way so that the
// -eq sums agree to the bit (checked entry by entry against the literals when this was written).
struct Prob {
	double p[256];
	Prob() {
		for(int q = 0; q < 256; ++q) {
			char b[80];
			snprintf(b, sizeof b, "%.32f", pow(10, -0.1 * q));
			p[q] = strtod(b, nullptr);
		}
	}
};
const Prob g_prob;

// growable byte buffer that is never zero-filled (a vector would touch every new page twice)
struct RawBuf {
	uint8_t *p = nullptr;
	size_t cap = 0;
	~RawBuf() { free(p); }
	uint8_t *data() const { return p; }
	size_t size() const { return cap; }
	void resize(size_t n) { if(n > cap) { p = (uint8_t *) realloc(p, n); cap = p ? n : 0; } }
	void release() { free(p); p = nullptr; cap = 0; }
};

// A stream of decompressed bytes with byte-level access (zlib reads plain files as they are)
struct Stream {
	gzFile f = nullptr;
	RawBuf buf;
	size_t pos = 0, end = 0;
	bool eof = false;
	bool pinned = false;       // while a batch is being located its bytes stay where they are (offsets remain valid)
	bool open(const char *path) {
		f = gzopen(path, "rb");
		if(!f) return false;
		gzbuffer(f, 1 << 20);
		buf.resize(8u << 20);
		return true;
	}
	void close() { if(f) gzclose(f); f = nullptr; }
	// make at least n bytes available at pos (fewer only at the end of the input)
	size_t ensure(size_t n) {
		if(end - pos >= n || eof) return end - pos;
		if(pos && !pinned) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
		n += pos;
		if(buf.size() < n) buf.resize(std::max(n, buf.size() * 2));
		while(end < n && !eof) {
			const int got = gzread(f, buf.data() + end, (unsigned) std::min<size_t>(buf.size() - end, 1u << 30));
			if(got <= 0) eof = true; else end += (size_t) got;
		}
		return end - pos;
	}
	// length of the line at pos without its '\n' (SIZE_MAX: the input ends before a newline)
	size_t line_len() {
		size_t have = end - pos, from = 0;
		for(;;) {
			const void *nl = memchr(buf.data() + pos + from, '\n', have - from);
			if(nl) return (size_t) ((const uint8_t *) nl - (buf.data() + pos));
			if(eof) return SIZE_MAX;
			from = have;
			have = ensure(have + (4u << 20));
			if(have == from) return SIZE_MAX;
		}
	}
	const uint8_t *at() const { return buf.data() + pos; }
};

struct Rec {
	std::string name;                 // header without the leading '@' / '>', chomped
	std::vector<uint8_t> seq, qual;   // codes (0-4, 8 for bytes outside the table) and raw quality bytes
};

// one FASTQ record located in the bytes of a chunk (which lives until the record has been packed)
struct Span {
	const uint8_t *name = nullptr, *seq = nullptr, *qual = nullptr;
	uint32_t name_len = 0, seq_len = 0;
	bool got = false;
};

// array that grows without being zero-filled and without a copy constructor in the way (mremap does the large moves)
template <class T> struct Arr {
	T *p = nullptr;
	size_t n = 0, cap = 0;
	Arr() = default;
	Arr(const Arr &) = delete;
	Arr &operator=(const Arr &) = delete;
	~Arr() { free(p); }
	T *data() const { return p; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	void clear() { n = 0; }
	void reserve(size_t m) { if(m > cap) { size_t c = std::max(m, cap + cap / 2); p = (T *) realloc(p, c * sizeof(T)); if(!p) abort(); cap = c; } }
	void resize_raw(size_t m) { reserve(m); n = m; }
	void push_back(const T &v) { if(n == cap) reserve(std::max<size_t>(1024, n + 1)); p[n++] = v; }
	void append(const T *q, size_t m) { if(m) { reserve(n + m); memcpy(p + n, q, m * sizeof(T)); n += m; } }
	T &operator[](size_t i) const { return p[i]; }
	T &back() const { return p[n - 1]; }
};

// what one worker thread makes of its share of the records; appended to the batch in thread order
struct alignas(128) Part {      // own cache lines: neighbouring threads push into neighbouring parts
	std::vector<uint8_t> codes[2];  // scratch
	std::vector<uint64_t> seq;
	std::vector<int32_t> len, N, nN;
	std::vector<char> names;
	std::vector<int32_t> name_len;
	std::vector<uint8_t> pair;
	int64_t records = 0;
	int max_len = 0;
	void clear() { seq.clear(); len.clear(); N.clear(); nN.clear(); names.clear(); name_len.clear(); pair.clear(); records = 0; max_len = 0; }
};

// A piece of the decompressed input. The bytes [lo, hi) of base are data; a streamed chunk keeps room in front of lo for
// the unfinished record of the chunk before it.
// Buffers of streamed chunks go round: the parser hands a chunk's buffer back when its records are packed, and the inflating threads
// take it again -- gigabytes of input pass through a few dozen buffers whose pages exist already, instead of through fresh memory that
// the kernel has to hand out page by page.
struct ChunkPool {
	std::mutex mu;
	std::vector<std::pair<uint8_t *, size_t>> spare;
	static constexpr size_t KEEP = 48;
	uint8_t *take(size_t cap) {
		{
			std::lock_guard<std::mutex> lk(mu);
			for(size_t i = 0; i < spare.size(); ++i) if(spare[i].second == cap) { uint8_t *p = spare[i].first; spare[i] = spare.back(); spare.pop_back(); return p; }
		}
		return (uint8_t *) malloc(cap ? cap : 1);
	}
	void give(uint8_t *p, size_t cap) {
		if(!p) return;
		{
			std::lock_guard<std::mutex> lk(mu);
			if(spare.size() < KEEP && cap >= (1u << 20)) { spare.push_back({p, cap}); return; }
		}
		free(p);
	}
	void clear() { std::lock_guard<std::mutex> lk(mu); for(auto &b : spare) free(b.first); spare.clear(); }
	~ChunkPool() { for(auto &b : spare) free(b.first); }
};
ChunkPool g_chunk_pool;

struct Chunk {
	uint8_t *base = nullptr;
	size_t cap = 0, lo = 0, hi = 0;
	bool mapped = false, last = false;
	bool io_error = false;          // the stream ended on a read / inflate error, not at its end
	size_t released = 0;            // mapped file: pages before this offset have been given back
	int64_t spans_end = 0;          // records located in it end here (position in the stream of records)
	~Chunk() { if(mapped) { if(base) munmap(base, cap); } else g_chunk_pool.give(base, cap); }
};

// The input file as chunks: a plain regular file is mapped whole (one chunk); a gzip file of several members (bgzip, concatenated
// lanes, `cat a.gz b.gz`) is mapped compressed and its members inflated side by side; anything else is read through zlib by a
// thread of its own that stays a few chunks ahead of the parser (so inflating overlaps with everything after it).
//
// Members in parallel: a gzip member carries no length, so where members start is guessed first -- every offset that looks
// like a member header (1f 8b 08, reserved flag bits clear, a plausible XFL and OS byte) -- and settled while inflating: offset
// 0 is a start; where the member that starts at a settled offset ENDS is the next settled start. A worker takes the candidates
// in ascending order, one member each, and inflates into chunks of its own; the parser is handed the chunks of the settled
// members in order and never sees what a worker made of a false candidate (which fails within bytes, being random data).
// A worker that is not feeding the parser stops at 512 MB of unread output until it is its member's turn, so at most that much
// per worker waits in memory. What the parser gets is byte for byte what gzread gives: members back to back, bytes behind the
// last member that are no member ignored, a corrupt member ending the input with an error after what came before it.
struct Feeder {
	size_t CHUNK = 32u << 20, HEAD = 1u << 20;      // (KMAHIP_INGEST_CHUNK: smaller ones for the tests)
	static constexpr size_t AHEAD = 4;
	gzFile gz = nullptr;
	Chunk *whole = nullptr;
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::deque<Chunk *> ready;
	bool done = false, stop = false;
	// members in parallel
	struct Seg { std::deque<Chunk *> ready; size_t buffered = 0; bool done = false, error = false; int64_t next = -1; };   // next: candidate where the member ended, -1: end of input
	const uint8_t *zmap = nullptr;
	size_t zsize = 0, LIMIT = 512u << 20;
	std::vector<size_t> cands;
	std::vector<Seg> segs;
	std::vector<std::thread> workers;
	size_t claim = 0, cur = 0;
	bool par = false, par_end = false;

	static bool looks_like_member(const uint8_t *q, size_t left) {
		return left >= 18 && q[0] == 0x1f && q[1] == 0x8b && q[2] == 8 && (q[3] & 0xe0) == 0 && (q[8] == 0 || q[8] == 2 || q[8] == 4) && (q[9] <= 13 || q[9] == 255);
	}
	void find_members(int nt) {
		std::vector<std::vector<size_t>> found((size_t) nt);
		auto scan = [&](int w) {
			const size_t a = zsize * (size_t) w / (size_t) nt, b = zsize * (size_t) (w + 1) / (size_t) nt;
			const uint8_t *q = zmap + a;
			while(q < zmap + b) {
				q = (const uint8_t *) memchr(q, 0x1f, (size_t) (zmap + b - q));
				if(!q) break;
				if(looks_like_member(q, (size_t) (zmap + zsize - q))) found[(size_t) w].push_back((size_t) (q - zmap));
				++q;
			}
		};
		std::vector<std::thread> pool;
		for(int w = 1; w < nt; ++w) pool.emplace_back(scan, w);
		scan(0);
		for(std::thread &t : pool) t.join();
		for(int w = 0; w < nt; ++w) cands.insert(cands.end(), found[(size_t) w].begin(), found[(size_t) w].end());
	}
	// one member, from candidate i
	void inflate_member(size_t i) {
		Seg &S = segs[i];
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		bool ok = inflateInit2(&zs, 15 + 16) == Z_OK, ended = false;
		size_t in_at = cands[i];
		Chunk *c = nullptr;
		auto hand_over = [&](Chunk *full) {
			std::unique_lock<std::mutex> lk(mu);
			S.ready.push_back(full);
			S.buffered += full->hi - full->lo;
			cv.notify_all();
			cv.wait(lk, [&] { return stop || i == cur || S.buffered < LIMIT; });
		};
		while(ok && !ended) {
			{ std::lock_guard<std::mutex> lk(mu); if(stop) break; }
			if(!c) {
				c = new Chunk();
				c->cap = HEAD + CHUNK;
				c->base = g_chunk_pool.take(c->cap);
				if(!c->base) { ok = false; break; }
				c->lo = c->hi = HEAD;
			}
			if(zs.avail_in == 0) {
				zs.next_in = const_cast<Bytef *>(zmap + in_at);
				zs.avail_in = (uInt) std::min<size_t>(zsize - in_at, 1u << 30);
				in_at += zs.avail_in;
			}
			zs.next_out = c->base + c->hi;
			zs.avail_out = (uInt) std::min<size_t>(c->cap - c->hi, 1u << 30);
			const uInt out0 = zs.avail_out;
			const int r = inflate(&zs, Z_NO_FLUSH);
			c->hi += out0 - zs.avail_out;
			if(r == Z_STREAM_END) ended = true;
			else if(r == Z_BUF_ERROR && zs.avail_in == 0 && in_at == zsize) ok = false;      // the file ends inside the member
			else if(r != Z_OK && r != Z_BUF_ERROR) ok = false;
			if(c->hi == c->cap && !ended && ok) { hand_over(c); c = nullptr; }
		}
		const size_t end = in_at - zs.avail_in;
		inflateEnd(&zs);
		int64_t next = -1;
		bool tail_error = false;
		if(ok && ended && end < zsize) {
			const auto it = std::lower_bound(cands.begin(), cands.end(), end);
			if(it != cands.end() && *it == end) next = (int64_t) (it - cands.begin());
			// else: bytes that are no member -- ignored, like gzread does. But bytes that begin like one (1f 8b) gzread would try
			// to inflate: a member whose header this reader's stricter test did not take for one (an unusual XFL / OS byte, fewer
			// than 18 bytes left) must not be dropped in silence -- the input ends here with a read error instead
			else if(zsize - end >= 2 && zmap[end] == 0x1f && zmap[end + 1] == 0x8b) tail_error = true;
		}
		std::lock_guard<std::mutex> lk(mu);
		// (what the chunk in hand holds of a member that failed is dropped, as gzread drops what a failing call had inflated: the
		// bytes next to the damage are not to be parsed)
		if(c && ok && ended) { S.ready.push_back(c); S.buffered += c->hi - c->lo; }
		else delete c;
		S.error = !(ok && ended) || tail_error; S.next = next; S.done = true;
		cv.notify_all();
	}
	void work() {
		for(;;) {
			size_t i;
			{
				std::lock_guard<std::mutex> lk(mu);
				if(stop || claim >= cands.size()) return;
				i = claim++;
			}
			inflate_member(i);
		}
	}

	bool open(const char *path) {
		const int fd = ::open(path, O_RDONLY);
		if(fd < 0) return false;
		if(const char *c = getenv("KMAHIP_INGEST_CHUNK")) { CHUNK = (size_t) std::max(64, atoi(c)); HEAD = std::max<size_t>(16, CHUNK / 32); }
		struct stat sb;
		uint8_t magic[2] = {0, 0};
		const bool regular = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
		if(regular && (sb.st_size < 2 || (pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)))) {
			whole = new Chunk();
			whole->last = true;
			if(sb.st_size > 0) {
				void *m = mmap(nullptr, (size_t) sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
				if(m == MAP_FAILED) { delete whole; whole = nullptr; }
				else {
					madvise(m, (size_t) sb.st_size, MADV_SEQUENTIAL);
					whole->base = (uint8_t *) m; whole->cap = (size_t) sb.st_size; whole->hi = (size_t) sb.st_size; whole->mapped = true;
				}
			}
			if(whole) { ::close(fd); return true; }
		}
		if(regular && sb.st_size >= 36 && !getenv("KMAHIP_INGEST_SERIAL_GZ")) {
			void *m = mmap(nullptr, (size_t) sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
			if(m != MAP_FAILED) {
				zmap = (const uint8_t *) m; zsize = (size_t) sb.st_size;
				const char *e = getenv("KMAHIP_INGEST_THREADS");
				const int hw = (int) std::thread::hardware_concurrency();
				int nt = e ? atoi(e) : std::min(16, hw > 0 ? hw : 1);
				nt = std::max(nt, 1);
				find_members((int) std::min<size_t>((size_t) nt, zsize / (1u << 20) + 1));
				if(cands.size() >= 2 && cands[0] == 0) {
					if(const char *l = getenv("KMAHIP_INGEST_GZ_LIMIT")) LIMIT = (size_t) std::max(1, atoi(l));
					par = true;
					segs.resize(cands.size());
					::close(fd);
					for(int w = 0; w < (int) std::min<size_t>((size_t) nt, cands.size()); ++w) workers.emplace_back([this] { work(); });
					return true;
				}
				munmap(m, zsize);
				zmap = nullptr; zsize = 0; cands.clear();
			}
		}
		gz = gzdopen(fd, "rb");
		if(!gz) { ::close(fd); return false; }
		gzbuffer(gz, 1 << 20);
		th = std::thread([this] { run(); });
		return true;
	}
	void run() {
		for(;;) {
			Chunk *c = new Chunk();
			c->cap = HEAD + CHUNK;
			c->base = g_chunk_pool.take(c->cap);
			c->lo = c->hi = HEAD;
			while(c->base && c->hi < c->cap) {
				// (in pieces: zlib drops what a call had inflated when the call ends in an error)
				const int got = gzread(gz, c->base + c->hi, (unsigned) std::min<size_t>(c->cap - c->hi, 1u << 20));
				if(got <= 0) {
					// (a file that ends inside a member reads as 0 bytes with Z_BUF_ERROR pending)
					int zerr = Z_OK;
					(void) gzerror(gz, &zerr);
					c->last = true; c->io_error = got < 0 || zerr == Z_BUF_ERROR || zerr == Z_DATA_ERROR;
					break;
				}
				c->hi += (size_t) got;
			}
			if(!c->base) c->last = true;
			const bool last = c->last;
			std::unique_lock<std::mutex> lk(mu);
			ready.push_back(c);
			if(last) done = true;
			cv.notify_all();
			if(last) return;
			cv.wait(lk, [this] { return ready.size() < AHEAD || stop; });
			if(stop) return;
		}
	}
	// the next chunk, nullptr after the last one
	Chunk *pop() {
		if(whole) { Chunk *c = whole; whole = nullptr; return c; }
		if(par) {
			std::unique_lock<std::mutex> lk(mu);
			for(;;) {
				if(par_end) return nullptr;
				Seg &S = segs[cur];
				cv.wait(lk, [&] { return !S.ready.empty() || S.done; });
				if(!S.ready.empty()) {
					Chunk *c = S.ready.front();
					S.ready.pop_front();
					S.buffered -= c->hi - c->lo;
					cv.notify_all();
					return c;
				}
				// the member is through: on to the one that starts where it ended, or the end of the input
				if(S.error || S.next < 0) {
					par_end = true;
					Chunk *c = new Chunk();
					c->last = true; c->io_error = S.error;
					return c;
				}
				cur = (size_t) S.next;
				cv.notify_all();
			}
		}
		if(!gz) return nullptr;
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [this] { return !ready.empty() || done; });
		if(ready.empty()) return nullptr;
		Chunk *c = ready.front();
		ready.pop_front();
		cv.notify_all();
		return c;
	}
	void close() {
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv.notify_all();
		if(th.joinable()) th.join();
		for(std::thread &w : workers) if(w.joinable()) w.join();
		workers.clear();
		for(Seg &S : segs) { for(Chunk *c : S.ready) delete c; S.ready.clear(); }
		if(zmap) munmap(const_cast<uint8_t *>(zmap), zsize);
		zmap = nullptr; par = false;
		for(Chunk *c : ready) delete c;
		ready.clear();
		delete whole; whole = nullptr;
		if(gz) gzclose(gz);
		gz = nullptr;
	}
};

// one mate file of a FASTQ input: its chunks, how far they have been cut into records, and the records waiting to be packed
struct Mate {
	Feeder feed;
	std::deque<Chunk *> live;      // chunks that located records still point into; back() is the one being cut
	size_t pos = 0;                 // offset in live.back(): everything before it has been located
	bool eof = false;               // no further record will come
	std::vector<Span> spans;        // located, not yet packed: [head, size)
	size_t head = 0;
	int64_t spans_base = 0;         // position of spans[0] in the stream of records
	~Mate() { for(Chunk *c : live) delete c; }
};

}  // namespace

struct kmahip_ingest {
	Stream s[2];                // FASTA input, and the first bytes of any input (format, phred scale)
	Mate m[2];                  // FASTQ input
	bool paired = false, fastq = true;
	bool interleaved = false;   // `-int file`: paired, the mates alternate in ONE file (run_input_INT, runinput.c:608-740); m[0] / s[0] only
	kmahip_trim trim;
	int phred = 33;
	int threads = 1;
	bool malformed = false, reported = false;      // a record that does not start with '@': everything before it is delivered, then KMAHIP_EFORMAT once
	bool io_error = false, io_reported = false;     // a corrupt .gz (or a failing read): what came before it is delivered, then KMAHIP_EIO once
	int64_t n_read = 0, n_kept = 0;
	int64_t batch_bases = 0;            // kmahip_ingest_set_batch_bases: a batch also closes once it holds about this many bases (0: no such bound)
	// current batch
	Arr<uint64_t> seq;
	Arr<int64_t> seq_off, N_off, name_off;
	Arr<int32_t> len, N;
	Arr<char> names;
	Arr<uint8_t> pair;
	// kept between batches: the worker threads' parts and the located records (steady state allocates nothing -- eight
	// threads growing fresh vectors every batch spent 6x the packing time inside the allocator)
	std::vector<Part> parts;
};

namespace {

// getPhredFileBuff, seqparse.c:551-589, over the first file buffer
int guess_phred(const uint8_t *buff0, size_t bytes) {
	long avail = (long) bytes;
	int scale = 33, maxlen = 0;
	const uint8_t *buff = buff0;
	while(avail) {
		int seek = 3;
		while(seek && --avail) if(*++buff == '\n') --seek;
		int len = 0;
		seek = avail ? 1 : 0;
		while(seek && --avail) {
			if(*++buff == '\n') seek = 0;
			else if(*buff < 33) return 0;
			else if(53 < *buff && *buff < 59) return 33;
			else if(94 < *buff) scale = 64;
			++len;
		}
		maxlen = std::max(maxlen, len);
	}
	return maxlen <= 301 ? scale : 33;
}

// FileBuffgetFq, seqparse.c:241-403, over bytes in memory: the record at p. `final`: the input ends at e.
//   LOC_REC        a record, *next = where the following one starts
//   LOC_MORE       the record runs past e and the input goes on (never with final)
//   LOC_END        the input ends here, or inside this record (which the reference drops)
//   LOC_MALFORMED  the record does not start with '@' ("Malformed input.", seqparse.c:256-260): the reference stops reading
// Every byte of the sequence line counts, also the '\r' of a DOS file (code 8): the reference's chomp loop stops at the newline
// code it has just stored (:322-326), so such a base is only lost later, to the quality trim ('\r' < '!').
enum { LOC_REC = 1, LOC_MORE = 0, LOC_END = 2, LOC_MALFORMED = -1 };
int locate_mem(const uint8_t *p, const uint8_t *e, bool final, Span &r, const uint8_t **next) {
	const int out = final ? LOC_END : LOC_MORE;
	if(p == e) return out;
	if(*p != '@') return LOC_MALFORMED;
	const uint8_t *nl = (const uint8_t *) memchr(p, '\n', (size_t) (e - p));
	if(!nl) return out;
	{	// header: everything up to the newline, chomped of trailing white space; the '@' is not part of the name
		const uint8_t *h = nl;
		while(h > p && isspace(h[-1])) --h;
		r.name = p + 1; r.name_len = h > p ? (uint32_t) (h - p - 1) : 0;
	}
	const uint8_t *q = nl + 1;
	nl = (const uint8_t *) memchr(q, '\n', (size_t) (e - q));
	if(!nl) return out;
	if((size_t) (nl - q) > (size_t) INT_MAX) return LOC_MALFORMED;
	r.seq = q; r.seq_len = (uint32_t) (nl - q);
	q = nl + 1;
	nl = (const uint8_t *) memchr(q, '\n', (size_t) (e - q));            // the '+' line
	if(!nl) return out;
	q = nl + 1;
	// quality: exactly as many raw bytes as the sequence line had, then on to the next newline
	if((size_t) (e - q) < r.seq_len) return out;
	r.qual = q;
	q += r.seq_len;
	nl = (const uint8_t *) memchr(q, '\n', (size_t) (e - q));
	r.got = true;
	if(!nl) {
		if(!final) return LOC_MORE;
		*next = e;                                                       // last record without a final newline
		return LOC_REC;
	}
	*next = nl + 1;
	return LOC_REC;
}

// records from p on until one starts at or after lim (or the bytes run out): appended to spans. Returns the status of the
// locate that stopped the loop (LOC_REC: stopped at lim), *stop = where it stopped.
int locate_range(const uint8_t *p, const uint8_t *lim, const uint8_t *e, bool final, std::vector<Span> &spans, const uint8_t **stop) {
	int st = LOC_REC;
	while(p < lim) {
		Span r;
		const uint8_t *next = nullptr;
		st = locate_mem(p, e, final, r, &next);
		if(st != LOC_REC) break;
		spans.push_back(r);
		p = next;
	}
	*stop = p;
	return st;
}

// A guess at the first record starting in [c, lim): a line that begins with '@' whose next line but one begins with '+'.
// (In a four-line FASTQ file only header lines are such lines; whether the guess was right is checked by the caller against
// where the records before it really end, so a wrong one costs time, never a different result.)
const uint8_t *guess_start(const uint8_t *lo, const uint8_t *c, const uint8_t *lim, const uint8_t *e) {
	const uint8_t *q = c;
	if(q > lo && q[-1] != '\n') {
		const uint8_t *nl = (const uint8_t *) memchr(q, '\n', (size_t) (e - q));
		if(!nl) return nullptr;
		q = nl + 1;
	}
	while(q < lim) {
		const uint8_t *l1 = (const uint8_t *) memchr(q, '\n', (size_t) (e - q));
		if(!l1) return nullptr;
		if(*q == '@') {
			const uint8_t *l2 = (const uint8_t *) memchr(l1 + 1, '\n', (size_t) (e - l1 - 1));
			if(!l2) return nullptr;
			if(l2 + 1 < e && l2[1] == '+') return q;
		}
		q = l1 + 1;
	}
	return nullptr;
}

// FileBuffgetFsa, seqparse.c:66-159: header line, then every byte the table knows up to the next '>'
bool next_fa(Stream &s, Rec &r) {
	r.seq.clear(); r.qual.clear(); r.name.clear();
	if(s.ensure(1) == 0) return false;
	size_t n = s.line_len();
	if(n == SIZE_MAX) return false;
	size_t e = n;
	while(e > 0 && isspace(s.at()[e - 1])) --e;
	r.name.assign((const char *) s.at() + 1, e > 0 ? e - 1 : 0);
	s.pos += n + 1;
	for(;;) {
		if(s.ensure(1) == 0) break;
		if(*s.at() == '>') break;
		const size_t have = s.end - s.pos;
		size_t i = 0;
		for(; i < have && s.at()[i] != '>'; ++i) {
			const uint8_t c = g_trans.t[s.at()[i]];
			if((c >> 3) == 0) r.seq.push_back(c);
		}
		s.pos += i;
	}
	return true;
}

// phredStat, runinput.c:127-313 (without a QC report). seq may be modified (hard masking). Returns the length figure the
// caller compares with minlen: end - start on the plain path, end - start - #N on the -eq / -mi path.
int phred_stat(uint8_t *seq, const uint8_t *qual, int len, const double *prob /* indexed by raw byte */, int minPhred /* raw */,
               int minQ, int hardmaskQ /* as given */, int minlen, int maxlen, int *START, int *END) {
	if(maxlen < len) { *START = 0; *END = 0; return 0; }
	int start = 0, end = len;
	while(start < end && qual[start] < minPhred) ++start;
	while(start < end && qual[end - 1] < minPhred) --end;
	len = end - start;
	if(!minQ && !hardmaskQ) { *START = start; *END = end; return len; }
	unsigned ns = 0, gc = 0;
	double sp = 0;
	for(int i = start; i < end; ++i) {
		sp += prob[qual[i]];
		if(seq[i] == 4 || qual[i] < hardmaskQ) { seq[i] = 4; ++ns; }
		else if(seq[i] == 1 || seq[i] == 2) ++gc;
	}
	const double minP = pow(10, (-0.1) * minQ);
	if(minlen <= (int) (len - ns) && (minP * len) < sp) {
		// 5' / 3' segments = a run of good bases followed by a run of bad ones, seen from the respective end
		unsigned ns5 = 0, ns3 = 0, l5 = 0, l3 = 0;
		double sp5 = 0, sp3 = 0;
		int p5 = start, p3 = end - 1;
		auto grow3 = [&]() {
			while((int) l3 < len && minPhred <= qual[p3]) { sp3 += prob[qual[p3]]; ++l3; if(seq[p3] == 4) ++ns3; --p3; }
			while((int) l3 < len && qual[p3] < minPhred) { sp3 += prob[qual[p3]]; ++l3; if(seq[p3] == 4) ++ns3; --p3; }
		};
		auto grow5 = [&]() {
			while((int) l5 < len && minPhred <= qual[p5]) { sp5 += prob[qual[p5]]; ++l5; if(seq[p5] == 4) ++ns5; ++p5; }
			while((int) l5 < len && qual[p5] < minPhred) { sp5 += prob[qual[p5]]; ++l5; if(seq[p5] == 4) ++ns5; ++p5; }
		};
		grow3();
		while(minlen <= (int) (len - ns) && (minP * len) < sp) {
			if((sp5 * l3) < (sp3 * l5)) {
				end -= (int) l3; ns -= ns3; len -= (int) l3; sp -= sp3;
				ns3 = 0; l3 = 0; sp3 = 0;
				grow3();
			} else {
				start += (int) l5; len -= (int) l5; ns -= ns5; sp -= sp5;
				ns5 = 0; l5 = 0; sp5 = 0;
				grow5();
			}
		}
	}
	(void) gc;
	*START = start; *END = end;
	return len - (int) ns;
}

// fsastat, runinput.c:315-368
int fsa_stat(const uint8_t *seq, int len, int maxlen, int *START, int *END) {
	if(maxlen < len) { *START = 0; *END = 0; return 0; }
	int start = 0, end = len;
	while(start < end && seq[end - 1] == 4) --end;
	while(start < end && seq[start] == 4) ++start;
	len = end - start;
	int ns = 0;
	for(int i = start + 1; i < end; ++i) if(seq[i] == 4) ++ns;        // the first base is never counted (:349-357)
	*START = start; *END = end;
	return len - ns;
}

// compDNA (compdna.c:99-127) appended to a part; codes outside 0-4 are OR-ed in unmasked, as the reference does
void append_read(Part &P, const uint8_t *codes, int L, const char *name, size_t name_len, uint8_t pair) {
	const size_t w0 = P.seq.size(), words = (size_t) ((L + 31) >> 5);
	P.seq.resize(w0 + words + 1, 0);                                   // + the pad word every read carries
	uint64_t *w = P.seq.data() + w0;
	int nN = 0;
	for(int i = 0; i < L; ++i) {
		const uint8_t c = codes[i];
		uint64_t &x = w[i >> 5];
		if(c == 4) { x <<= 2; P.N.push_back(i); ++nN; }
		else x = (x << 2) | c;
	}
	if(L & 31) w[words - 1] <<= (64 - ((L & 31) << 1));
	P.len.push_back(L); P.nN.push_back(nN);
	P.names.insert(P.names.end(), name, name + name_len);
	P.names.push_back('\0');
	P.name_len.push_back((int32_t) name_len + 1);
	P.pair.push_back(pair);
	P.max_len = std::max(P.max_len, L);
}

// the same straight from the raw bytes of the sequence line (translation fused into the packing loop)
void append_raw(Part &P, const uint8_t *raw, int L, const char *name, size_t name_len, uint8_t pair) {
	const size_t w0 = P.seq.size(), words = (size_t) ((L + 31) >> 5);
	P.seq.resize(w0 + words + 1, 0);
	uint64_t *w = P.seq.data() + w0;
	int nN = 0;
	for(int i0 = 0; i0 < L; i0 += 32) {
		const int e = std::min(32, L - i0);
		uint64_t x = 0;
		for(int i = 0; i < e; ++i) {
			const uint8_t c = g_trans.t[raw[i0 + i]];
			if(c == 4) { x <<= 2; P.N.push_back(i0 + i); ++nN; }
			else x = (x << 2) | c;
		}
		w[i0 >> 5] = e < 32 ? x << (64 - (e << 1)) : x;
	}
	P.len.push_back(L); P.nN.push_back(nN);
	P.names.insert(P.names.end(), name, name + name_len);
	P.names.push_back('\0');
	P.name_len.push_back((int32_t) name_len + 1);
	P.pair.push_back(pair);
	P.max_len = std::max(P.max_len, L);
}

// trim, gate and pack records [a, b) of a located batch (run_input / run_input_PE, runinput.c:404-424, 515-549)
void pack_fastq(const kmahip_ingest *in, const Span *const *spans, size_t a, size_t b, Part &P) {
	P.clear();
	const kmahip_trim &T = in->trim;
	const double *prob = g_prob.p - in->phred;          // indexed by the raw quality byte
	const int mates = in->paired ? 2 : 1;
	std::vector<uint8_t> *codes = P.codes;
	const bool plain = !T.min_q && !T.hardmask_q;       // only the ends are trimmed, by quality alone (runinput.c:144-171)
	const bool il = in->interleaved;                     // (mate m of record i is record 2 i + m of the one file)
	auto span_of = [&](int m, size_t i) -> const Span & { return il ? spans[0][2 * i + (size_t) m] : spans[m][i]; };
	for(size_t i = a; i < b; ++i) {
		int st[2] = {0, 0}, en[2] = {0, 0}, len[2] = {0, 0};
		if(plain) {
			const int minPhred = in->phred + T.min_phred;
			for(int m = 0; m < mates; ++m) {
				const Span &r = span_of(m, i);
				const uint8_t *q = r.qual;
				const int L = r.got ? (int) r.seq_len : 0;
				if(T.max_len < L) continue;
				int s0 = 0, e0 = L;
				while(s0 < e0 && q[s0] < minPhred) ++s0;
				while(s0 < e0 && q[e0 - 1] < minPhred) --e0;
				st[m] = s0; en[m] = e0; len[m] = e0 - s0;
			}
			const bool ok0 = T.min_len <= len[0], ok1 = in->paired && T.min_len <= len[1];
			auto put = [&](int m, uint8_t pair) {
				const Span &r = span_of(m, i);
				append_raw(P, r.seq + st[m], en[m] - st[m], (const char *) r.name, r.got ? r.name_len : 0, pair);
			};
			if(ok0 && ok1) { put(0, 1); put(1, 2); }
			else if(ok0) put(0, 0);
			else if(ok1) put(1, 0);
			else continue;
			++P.records;
			continue;
		}
		for(int m = 0; m < mates; ++m) {
			const Span &r = span_of(m, i);
			const int L = r.got ? (int) r.seq_len : 0;       // a mate file that ran out yields empty mates (the `|` at :516)
			codes[m].resize((size_t) L);
			for(int x = 0; x < L; ++x) codes[m][(size_t) x] = g_trans.t[r.seq[x]];
			len[m] = phred_stat(codes[m].data(), r.qual, L, prob, in->phred + T.min_phred, T.min_q, T.hardmask_q, T.min_len,
			                    T.max_len, &st[m], &en[m]);
		}
		const bool ok0 = T.min_len <= len[0], ok1 = in->paired && T.min_len <= len[1];
		auto put = [&](int m, uint8_t pair) {
			const Span &r = span_of(m, i);
			append_read(P, codes[m].data() + st[m], en[m] - st[m], (const char *) r.name, r.got ? r.name_len : 0, pair);
		};
		if(ok0 && ok1) { put(0, 1); put(1, 2); }
		else if(ok0) put(0, 0);
		else if(ok1) put(1, 0);
		else continue;
		++P.records;
	}
}

void append_part(kmahip_ingest *in, const Part &P) {
	const size_t r0 = in->len.size();
	in->seq.append(P.seq.data(), P.seq.size());
	in->len.append(P.len.data(), P.len.size());
	in->N.append(P.N.data(), P.N.size());
	in->names.append(P.names.data(), P.names.size());
	in->pair.append(P.pair.data(), P.pair.size());
	for(size_t i = 0; i < P.len.size(); ++i) {
		in->seq_off.push_back(in->seq_off[r0 + i] + ((P.len[i] + 31) >> 5) + 1);
		in->N_off.push_back(in->N_off[r0 + i] + P.nN[i]);
		in->name_off.push_back(in->name_off[r0 + i] + P.name_len[i]);
	}
}

// the parts of one round of packing appended to the batch, each by the thread that made it
void append_parts(kmahip_ingest *in, const std::vector<Part> &parts, int nt) {
	std::vector<size_t> r0((size_t) nt + 1), w0((size_t) nt + 1), n0((size_t) nt + 1), c0((size_t) nt + 1);
	r0[0] = in->len.size(); w0[0] = in->seq.size(); n0[0] = in->N.size(); c0[0] = in->names.size();
	for(int t = 0; t < nt; ++t) {
		const Part &P = parts[(size_t) t];
		r0[(size_t) t + 1] = r0[(size_t) t] + P.len.size(); w0[(size_t) t + 1] = w0[(size_t) t] + P.seq.size();
		n0[(size_t) t + 1] = n0[(size_t) t] + P.N.size(); c0[(size_t) t + 1] = c0[(size_t) t] + P.names.size();
	}
	in->len.resize_raw(r0[(size_t) nt]); in->pair.resize_raw(r0[(size_t) nt]);
	in->seq_off.resize_raw(r0[(size_t) nt] + 1); in->N_off.resize_raw(r0[(size_t) nt] + 1); in->name_off.resize_raw(r0[(size_t) nt] + 1);
	in->seq.resize_raw(w0[(size_t) nt]); in->N.resize_raw(n0[(size_t) nt]); in->names.resize_raw(c0[(size_t) nt]);
	auto copy = [&](int t) {
		const Part &P = parts[(size_t) t];
		const size_t r = r0[(size_t) t];
		if(!P.seq.empty()) memcpy(in->seq.data() + w0[(size_t) t], P.seq.data(), P.seq.size() * sizeof(uint64_t));
		if(!P.len.empty()) memcpy(in->len.data() + r, P.len.data(), P.len.size() * sizeof(int32_t));
		if(!P.N.empty()) memcpy(in->N.data() + n0[(size_t) t], P.N.data(), P.N.size() * sizeof(int32_t));
		if(!P.names.empty()) memcpy(in->names.data() + c0[(size_t) t], P.names.data(), P.names.size());
		if(!P.pair.empty()) memcpy(in->pair.data() + r, P.pair.data(), P.pair.size());
		int64_t so = (int64_t) w0[(size_t) t], no = (int64_t) n0[(size_t) t], co = (int64_t) c0[(size_t) t];
		for(size_t i = 0; i < P.len.size(); ++i) {
			so += ((P.len[i] + 31) >> 5) + 1; no += P.nN[i]; co += P.name_len[i];
			in->seq_off[r + i + 1] = so; in->N_off[r + i + 1] = no; in->name_off[r + i + 1] = co;
		}
	};
	std::vector<std::thread> pool;
	for(int t = 1; t < nt; ++t) pool.emplace_back(copy, t);
	copy(0);
	for(std::thread &th : pool) th.join();
}

// Cut more of a mate file into records: one wave of bytes, its regions located by `threads` threads side by side and
// stitched together in order. A region's records are taken as they are when the region began exactly where the records
// before it end; otherwise that stretch is located again from the true position, one record after the other -- the result is
// always that of reading the file front to back. Returns false when nothing more will come.
bool fill_wave(kmahip_ingest *in, Mate &M) {
	// (KMAHIP_INGEST_REGION: the least bytes a region takes, 256 KiB; tiny ones make small test files take every path here)
	const size_t region_min = getenv("KMAHIP_INGEST_REGION") ? (size_t) std::max(16, atoi(getenv("KMAHIP_INGEST_REGION"))) : (256u << 10);
	size_t wave = (size_t) in->threads * std::min<size_t>(8u << 20, region_min * 32);
	struct Region { const uint8_t *lim = nullptr, *start = nullptr, *stop = nullptr; int st = LOC_REC; std::vector<Span> spans; };
	while(!M.eof) {
		if(M.live.empty() || (M.live.back()->hi - M.pos < std::min<size_t>(64u << 10, region_min) && !M.live.back()->last)) {
			// the next chunk, with what is left of this one (an unfinished record) in front of it
			Chunk *c = M.feed.pop();
			Chunk *old = M.live.empty() ? nullptr : M.live.back();
			const size_t tail = old ? old->hi - M.pos : 0;
			if(c && c->io_error) in->io_error = true;
			if(!c) {
				if(!old) { M.eof = true; break; }
				old->last = true;                              // (a stream that ended on a chunk boundary)
			} else {
				if(tail > c->lo) {                              // a record longer than the room kept for it
					Chunk *big = new Chunk();
					big->cap = tail + (c->hi - c->lo);
					big->base = (uint8_t *) malloc(big->cap ? big->cap : 1);
					if(!big->base) abort();
					memcpy(big->base + tail, c->base + c->lo, c->hi - c->lo);
					big->lo = tail; big->hi = big->cap; big->last = c->last;
					delete c;
					c = big;
				}
				if(tail) memcpy(c->base + c->lo - tail, old->base + M.pos, tail);
				c->lo -= tail;
				if(old) old->spans_end = M.spans_base + (int64_t) M.spans.size();
				M.live.push_back(c);
				M.pos = c->lo;
			}
		}
		Chunk *c = M.live.back();
		const uint8_t *lo = c->base + M.pos, *end = c->base + c->hi;
		const uint8_t *e = (size_t) (end - lo) > wave ? lo + wave : end;
		const bool final = e == end && c->last;
		if(lo == e) { if(final) M.eof = true; continue; }
		const int nr = (int) std::max<size_t>(1, std::min<size_t>((size_t) in->threads, (size_t) (e - lo) / region_min));
		std::vector<Region> R((size_t) nr);
		auto work = [&](int k) {
			Region &r = R[(size_t) k];
			const uint8_t *c0 = lo + (size_t) (e - lo) * (size_t) k / (size_t) nr;
			r.lim = k + 1 == nr ? e : lo + (size_t) (e - lo) * (size_t) (k + 1) / (size_t) nr;
			r.start = k == 0 ? lo : guess_start(lo, c0, r.lim, e);
			if(r.start) r.st = locate_range(r.start, r.lim, e, final, r.spans, &r.stop);
		};
		{
			std::vector<std::thread> pool;
			for(int k = 1; k < nr; ++k) pool.emplace_back(work, k);
			work(0);
			for(std::thread &th : pool) th.join();
		}
		const uint8_t *cur = lo;
		int st = LOC_REC;
		const size_t before = M.spans.size();
		for(int k = 0; k < nr && st == LOC_REC; ++k) {
			Region &r = R[(size_t) k];
			if(cur >= r.lim) continue;                          // a record before it reaches over the whole region
			if(r.start != cur) {
				// not where the guess was (or no guess): from the true position up to the guessed start, or through the region
				const uint8_t *to = r.start && r.start > cur ? r.start : r.lim;
				st = locate_range(cur, to, e, final, M.spans, &cur);
				if(st != LOC_REC) break;
				if(cur != r.start) {
					if(cur < r.lim) st = locate_range(cur, r.lim, e, final, M.spans, &cur);
					continue;
				}
			}
			M.spans.insert(M.spans.end(), r.spans.begin(), r.spans.end());
			cur = r.stop; st = r.st;
		}
		M.pos = (size_t) (cur - c->base);
		if(st == LOC_MALFORMED) { in->malformed = true; M.eof = true; }
		else if(st == LOC_END) M.eof = true;
		else if(final && cur == e) M.eof = true;
		if(M.spans.size() > before) return true;
		if(M.eof) break;
		// no record fits into the bytes looked at: look at more of the chunk, or go on to the next one
		if(e < end) wave *= 2;
		else if(c->last) M.eof = true;
		else {
			// force the next chunk in, whatever the size of the tail
			Chunk *n = M.feed.pop();
			const size_t tail = c->hi - M.pos;
			if(n && n->io_error) in->io_error = true;
			if(!n) { c->last = true; continue; }
			Chunk *big = new Chunk();
			big->cap = tail + (n->hi - n->lo);
			big->base = (uint8_t *) malloc(big->cap ? big->cap : 1);
			if(!big->base) abort();
			memcpy(big->base, c->base + M.pos, tail);
			memcpy(big->base + tail, n->base + n->lo, n->hi - n->lo);
			big->hi = big->cap; big->last = n->last;
			delete n;
			c->spans_end = M.spans_base + (int64_t) M.spans.size();
			M.live.push_back(big);
			M.pos = 0;
		}
	}
	return false;
}

// chunks whose records have all been packed go
void release_chunks(Mate &M) {
	const int64_t done = M.spans_base + (int64_t) M.head;
	while(M.live.size() > 1 && M.live.front()->spans_end <= done) { delete M.live.front(); M.live.pop_front(); }
	// a file that is mapped whole: the pages behind the records packed so far are given back, so that what the process holds of
	// its input does not grow with the input (they stay in the page cache; nothing here reads them again)
	if(M.live.size() == 1 && M.live.front()->mapped && M.live.front()->base) {
		Chunk *c = M.live.front();
		const uint8_t *upto = M.head < M.spans.size() && M.spans[M.head].name ? M.spans[M.head].name - 1 : c->base + M.pos;
		const size_t page = 4096, from = (c->released + page - 1) & ~(page - 1);
		const size_t to = ((size_t) (upto - c->base)) & ~(page - 1);
		if(to > from + (8u << 20)) { madvise(c->base + from, to - from, MADV_DONTNEED); c->released = to; }
	}
	if(M.head == M.spans.size()) { M.spans_base += (int64_t) M.head; M.spans.clear(); M.head = 0; }
}

}  // namespace

extern "C" void kmahip_trim_default(kmahip_trim *t) {
	if(!t) return;
	t->min_phred = 20; t->min_q = 0; t->hardmask_q = 0; t->min_len = 16; t->max_len = INT_MAX;
}

static int ingest_open_impl(const char *path1, const char *path2, bool interleaved, const kmahip_trim *trim, kmahip_ingest **out);

extern "C" int kmahip_ingest_open(const char *path1, const char *path2, const kmahip_trim *trim, kmahip_ingest **out) {
	return ingest_open_impl(path1, path2, false, trim, out);
}

// `-int file` (run_input_INT, runinput.c:608-740): the records of ONE file taken two at a time as the mates of a couple, trimmed and
// gated like the two files of -ipe; a last record without a partner meets an empty mate (FileBuffgetFq sets qseq->len = 0 before it
// finds the end of the file, seqparse.c:249) and is filed singly. FASTQ only: for FASTA the reference cuts mate 2 with mate 1's
// bounds (runinput.c:705: `qseq2->len = end - start`), which may reach past what it read.
extern "C" int kmahip_ingest_open_interleaved(const char *path, const kmahip_trim *trim, kmahip_ingest **out) {
	int rc = ingest_open_impl(path, nullptr, true, trim, out);
	if(!rc && !(*out)->fastq) {
		kmahip_ingest_close(*out); *out = nullptr;
		kmahip_set_error("%s: interleaved input is taken as FASTQ only", path);
		return KMAHIP_EFORMAT;
	}
	return rc;
}

static int ingest_open_impl(const char *path1, const char *path2, bool interleaved, const kmahip_trim *trim, kmahip_ingest **out) {
	if(!path1 || !out || (interleaved && path2)) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	kmahip_ingest *in = new kmahip_ingest();
	if(trim) in->trim = *trim; else kmahip_trim_default(&in->trim);
	// kma.c:1555-1557, runinput.c:380-382
	if(in->trim.min_phred < in->trim.hardmask_q) in->trim.min_phred = in->trim.hardmask_q;
	if(in->trim.min_phred < in->trim.min_q) in->trim.min_phred = in->trim.min_q;
	in->paired = path2 != nullptr || interleaved;
	in->interleaved = interleaved;
	const int files = path2 ? 2 : 1;
	{	// worker threads for trimming + packing: KMAHIP_INGEST_THREADS, else the hardware threads, at most 16
		const char *e = getenv("KMAHIP_INGEST_THREADS");
		const int hw = (int) std::thread::hardware_concurrency();
		in->threads = e ? atoi(e) : std::min(16, hw > 0 ? hw : 1);
		if(in->threads < 1) in->threads = 1;
	}
	const char *paths[2] = {path1, path2};
	int kind[2] = {0, 0};
	for(int i = 0; i < files; ++i) {
		if(!in->s[i].open(paths[i])) { kmahip_set_error("cannot open %s", paths[i]); kmahip_ingest_close(in); return KMAHIP_EIO; }
		const size_t have = in->s[i].ensure(FIRST_CHUNK);
		const uint8_t c = have ? *in->s[i].at() : 0;
		kind[i] = c == '@' ? 1 : (c == '>' ? 2 : 0);
		if(have && !kind[i]) { kmahip_set_error("cannot determine format of file %s", paths[i]); kmahip_ingest_close(in); return KMAHIP_EFORMAT; }
	}
	if(files == 2 && kind[0] != kind[1]) { kmahip_set_error("%s and %s are in different formats", path1, path2); kmahip_ingest_close(in); return KMAHIP_EFORMAT; }
	in->fastq = kind[0] != 2;
	if(kind[0] == 1) {
		in->phred = guess_phred(in->s[0].at(), std::min(in->s[0].end - in->s[0].pos, FIRST_CHUNK));
		if(files == 2 && in->phred == 0) in->phred = guess_phred(in->s[1].at(), std::min(in->s[1].end - in->s[1].pos, FIRST_CHUNK));
	}
	if(in->fastq) {
		// FASTQ is read again from the start, in chunks: a mapped file or a thread that inflates ahead of the parser
		for(int i = 0; i < files; ++i) {
			in->s[i].close();
			in->s[i].buf.release();
			if(!in->m[i].feed.open(paths[i])) { kmahip_set_error("cannot open %s", paths[i]); kmahip_ingest_close(in); return KMAHIP_EIO; }
		}
	}
	*out = in;
	return KMAHIP_OK;
}

// One rank's part of the input (see kmahip.h). Plain single-end FASTQ: the mapped file is cut at record starts -- the guess of
// fill_wave's regions, taken from the byte part * size / parts onwards, the same on every rank -- and the reader is given
// [start of this part, start of the next); everything else is delivered whole for the caller to slice by record number.
extern "C" int kmahip_ingest_open_part(const char *path1, const char *path2, const kmahip_trim *trim, int part, int parts, kmahip_ingest **out,
                                       int *whole_input) {
	if(!whole_input || parts < 1 || part < 0 || part >= parts) { kmahip_set_error("bad part"); return KMAHIP_EINVAL; }
	*whole_input = 1;
	// (path2 == "" asks for the interleaved reader: the records of path1 two at a time, kmahip_ingest_open_interleaved)
	int rc = (path2 && !*path2) ? kmahip_ingest_open_interleaved(path1, trim, out) : kmahip_ingest_open(path1, path2, trim, out);
	if(rc || parts == 1) { if(!rc) *whole_input = 0; return rc; }
	kmahip_ingest *in = *out;
	Chunk *w = (in->fastq && !in->paired) ? in->m[0].feed.whole : nullptr;
	if(!w || !w->mapped || !w->base || getenv("KMAHIP_INGEST_NO_RANGES")) return KMAHIP_OK;
	const uint8_t *base = w->base, *end = w->base + w->hi;
	auto cut = [&](int q) -> size_t {
		if(q <= 0) return 0;
		if(q >= parts) return w->hi;
		const uint8_t *g = guess_start(base, base + (size_t) ((unsigned __int128) w->hi * (unsigned) q / (unsigned) parts), end, end);
		return g ? (size_t) (g - base) : w->hi;
	};
	const size_t a = cut(part), b = cut(part + 1);
	w->lo = a; w->hi = b < a ? a : b;
	*whole_input = 0;
	return KMAHIP_OK;
}

extern "C" int kmahip_ingest_next(kmahip_ingest *in, int64_t max_records, kmahip_read_batch *batch) {
	if(!in || !batch || max_records < 0) { kmahip_set_error("bad argument"); return KMAHIP_EINVAL; }
	in->seq.clear(); in->len.clear(); in->N.clear(); in->names.clear(); in->pair.clear();
	in->seq_off.clear(); in->N_off.clear(); in->name_off.clear();
	in->seq_off.push_back(0); in->N_off.push_back(0); in->name_off.push_back(0);
	const kmahip_trim &T = in->trim;
	const bool il = in->interleaved;
	const int mates = (in->paired && !il) ? 2 : 1;          // mate FILES
	int64_t records = 0;
	int max_len = 0;
	if(in->fastq) {
		// Records are located a wave of bytes at a time (fill_wave), then trimmed and packed by the same few threads and
		// appended to the batch; a batch takes in as many input records as it still has room for, until it holds
		// `max_records` kept ones or the input ends.
		const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
		double ms_locate = 0, ms_pack = 0, ms_gather = 0;
		auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
		while(records < max_records && (in->batch_bases <= 0 || (int64_t) in->seq.size() * 32 < in->batch_bases)) {
			const auto t0 = std::chrono::steady_clock::now();
			size_t avail = 0;
			for(;;) {
				for(int m = 0; m < mates; ++m) {
					Mate &M = in->m[m], &O = in->m[mates - 1 - m];
					// a mate file that has run out yields empty mates for what the other still holds (the `|` at runinput.c:516)
					if(M.eof && M.spans.size() - M.head < O.spans.size() - O.head) M.spans.resize(M.head + (O.spans.size() - O.head));
				}
				avail = in->m[0].spans.size() - in->m[0].head;
				if(mates == 2) avail = std::min(avail, in->m[1].spans.size() - in->m[1].head);
				if(il) {          // couples of consecutive records; the last record of an odd file meets an empty mate
					if(avail == 1 && in->m[0].eof) in->m[0].spans.resize(in->m[0].head + 2);
					avail = (in->m[0].spans.size() - in->m[0].head) / 2;
				}
				if(avail) break;
				bool more = false;
				for(int m = 0; m < mates; ++m) if(in->m[m].spans.size() - in->m[m].head < (il ? 2u : 1u)) more |= fill_wave(in, in->m[m]);
				if(!more) {
					bool pending = false;
					for(int m = 0; m < mates; ++m) pending |= in->m[m].spans.size() > in->m[m].head;
					if(!pending) break;
				}
			}
			if(!avail) break;
			const size_t n = (size_t) std::min<int64_t>(max_records - records, (int64_t) avail);
			in->n_read += (int64_t) n;
			const auto t1 = std::chrono::steady_clock::now();
			const Span *spans[2] = {in->m[0].spans.data() + in->m[0].head, mates == 2 ? in->m[1].spans.data() + in->m[1].head : nullptr};
			const int nt = (int) std::max<size_t>(1, std::min<size_t>((size_t) in->threads, n / 2048));
			if(in->parts.size() < (size_t) nt) in->parts.resize((size_t) nt);
			std::vector<Part> &parts = in->parts;
			{
				std::vector<std::thread> pool;
				for(int t = 1; t < nt; ++t) pool.emplace_back(pack_fastq, in, spans, n * (size_t) t / (size_t) nt, n * (size_t) (t + 1) / (size_t) nt, std::ref(parts[(size_t) t]));
				pack_fastq(in, spans, 0, n / (size_t) nt, parts[0]);
				for(std::thread &th : pool) th.join();
			}
			const auto t2 = std::chrono::steady_clock::now();
			append_parts(in, parts, nt);
			for(int t = 0; t < nt; ++t) { records += parts[(size_t) t].records; max_len = std::max(max_len, parts[(size_t) t].max_len); }
			for(int m = 0; m < mates; ++m) { in->m[m].head += il ? 2 * n : n; release_chunks(in->m[m]); }
			const auto t3 = std::chrono::steady_clock::now();
			ms_locate += ms(t0, t1); ms_pack += ms(t1, t2); ms_gather += ms(t2, t3);
		}
		if(dbg) fprintf(stderr, "[kmahip] ingest: %lld records kept (%d threads): locate %.1f ms, trim + pack %.1f ms, gather %.1f ms\n",
		                (long long) records, in->threads, ms_locate, ms_pack, ms_gather);
	} else {
		Rec r[2];
		Part P;
		while(records < max_records && (in->batch_bases <= 0 || (int64_t) P.seq.size() * 32 < in->batch_bases)) {
			bool got[2] = {false, false};
			for(int m = 0; m < mates; ++m) got[m] = next_fa(in->s[m], r[m]);
			if(!got[0] && !got[1]) break;
			++in->n_read;
			int st[2] = {0, 0}, en[2] = {0, 0}, len[2] = {0, 0};
			for(int m = 0; m < mates; ++m) {
				if(!got[m]) { r[m].seq.clear(); r[m].name.clear(); }
				len[m] = fsa_stat(r[m].seq.data(), (int) r[m].seq.size(), T.max_len, &st[m], &en[m]);
			}
			const bool ok0 = T.min_len <= len[0], ok1 = in->paired && T.min_len <= len[1];
			auto put = [&](int m, uint8_t pair) { append_read(P, r[m].seq.data() + st[m], en[m] - st[m], r[m].name.data(), r[m].name.size(), pair); };
			if(ok0 && ok1) { put(0, 1); put(1, 2); }
			else if(ok0) put(0, 0);
			else if(ok1) put(1, 0);
			else continue;
			++records;
		}
		append_part(in, P);
		max_len = P.max_len;
	}
	in->n_kept += records;
	if(in->N.empty()) in->N.push_back(0);                  // never hand out a null pointer
	memset(batch, 0, sizeof *batch);
	batch->reads.n_reads = (int64_t) in->len.size();
	batch->reads.seq = in->seq.data(); batch->reads.seq_off = in->seq_off.data(); batch->reads.len = in->len.data();
	batch->reads.N = in->N.data(); batch->reads.N_off = in->N_off.data();
	batch->reads.seq_words = (int64_t) in->seq.size(); batch->reads.N_total = in->N_off.back();
	batch->reads.max_len = max_len;
	in->names.reserve(1); in->pair.reserve(1);          // (an empty batch still hands out pointers)
	batch->names = in->names.data(); batch->name_off = in->name_off.data(); batch->pair = in->pair.data();
	batch->records = records;
	if(in->io_error && !in->io_reported && records == 0) {
		in->io_reported = true;
		kmahip_set_error("read error (corrupt or truncated compressed input) after %lld records", (long long) in->n_read);
		return KMAHIP_EIO;
	}
	if(in->malformed && !in->reported && records == 0) {
		// like the reference, which prints "Malformed input." and ends with a non-zero exit status after the good records
		in->reported = true;
		kmahip_set_error("malformed FASTQ input after %lld records", (long long) in->n_read);
		return KMAHIP_EFORMAT;
	}
	return KMAHIP_OK;
}

// A second bound on a batch beside kmahip_ingest_next's max_records: once the batch holds about max_bases bases (counted in packed
// words, checked between the waves of located records: a batch may go over by one wave) it closes. For inputs of long reads, where a
// count of reads says little about the size of a batch. 0: no such bound.
extern "C" int kmahip_ingest_set_batch_bases(kmahip_ingest *in, int64_t max_bases) {
	if(!in || max_bases < 0) { kmahip_set_error("bad argument"); return KMAHIP_EINVAL; }
	in->batch_bases = max_bases;
	return KMAHIP_OK;
}

// what a caller that takes the whole input as ONE batch has to ask afterwards: did the input break off behind the records delivered?
// (kmahip_ingest_next reports that with the call that delivers nothing -- which such a caller never makes, and must not make while
// it still uses the batch: the next call reuses the batch's arrays.)
extern "C" int kmahip_ingest_status(kmahip_ingest *in) {
	if(!in) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(in->io_error && !in->io_reported) {
		in->io_reported = true;
		kmahip_set_error("read error (corrupt or truncated compressed input) after %lld records", (long long) in->n_read);
		return KMAHIP_EIO;
	}
	if(in->malformed && !in->reported) {
		in->reported = true;
		kmahip_set_error("malformed FASTQ input after %lld records", (long long) in->n_read);
		return KMAHIP_EFORMAT;
	}
	return KMAHIP_OK;
}

extern "C" int kmahip_ingest_phred_scale(const kmahip_ingest *in) { return in ? in->phred : 0; }

extern "C" void kmahip_ingest_counts(const kmahip_ingest *in, int64_t *records_read, int64_t *records_kept) {
	if(records_read) *records_read = in ? in->n_read : 0;
	if(records_kept) *records_kept = in ? in->n_kept : 0;
}

extern "C" void kmahip_ingest_close(kmahip_ingest *in) {
	if(!in) return;
	in->s[0].close(); in->s[1].close();
	in->m[0].feed.close(); in->m[1].feed.close();
	for(int m = 0; m < 2; ++m) { for(Chunk *c : in->m[m].live) delete c; in->m[m].live.clear(); }
	g_chunk_pool.clear();
	delete in;
}
