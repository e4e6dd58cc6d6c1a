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
#include <cstring>
#include <string>
#include <thread>
#include <vector>

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

// one FASTQ record located in a stream buffer (offsets from the start of the buffer)
struct Span {
	size_t name = 0, name_len = 0, seq = 0, seq_len = 0, qual = 0;
	bool got = false;
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

}  // namespace

struct kmahip_ingest {
	Stream s[2];
	bool paired = false, fastq = true;
	kmahip_trim trim;
	int phred = 33;
	int threads = 1;
	bool malformed = false, reported = false;      // a record that does not start with '@': everything before it is delivered, then KMAHIP_EFORMAT once
	int64_t n_read = 0, n_kept = 0;
	// current batch
	std::vector<uint64_t> seq;
	std::vector<int64_t> seq_off, N_off, name_off;
	std::vector<int32_t> len, N;
	std::vector<char> names;
	std::vector<uint8_t> pair;
	// kept between batches: the worker threads' parts and the located records (steady state allocates nothing -- eight
	// threads growing fresh vectors every batch spent 6x the packing time inside the allocator)
	std::vector<Part> parts;
	std::vector<Span> spans[2];
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

// FileBuffgetFq, seqparse.c:241-403: locate the next record. false at the end of the input (or on a truncated / malformed
// record). Every byte of the sequence line counts, also the '\r' of a DOS file (code 8): the reference's chomp loop stops at
// the newline code it has just stored (:322-326), so such a base is only lost later, to the quality trim ('\r' < '!').
bool locate_fq(Stream &s, Span &r, bool *malformed) {
	r = Span();
	if(s.ensure(1) == 0) return false;
	if(*s.at() != '@') { *malformed = true; return false; }       // "Malformed input." (seqparse.c:256-260): the reference stops reading here
	size_t n = s.line_len();
	if(n == SIZE_MAX) return false;
	{	// header: everything up to the newline, chomped of trailing white space; the '@' is not part of the name
		size_t e = n;
		while(e > 0 && isspace(s.at()[e - 1])) --e;
		r.name = s.pos + 1; r.name_len = e > 0 ? e - 1 : 0;
		s.pos += n + 1;
	}
	s.ensure(1);
	n = s.line_len();
	if(n == SIZE_MAX) return false;
	r.seq = s.pos; r.seq_len = n;
	s.pos += n + 1;
	s.ensure(1);
	n = s.line_len();                                                   // the '+' line
	if(n == SIZE_MAX) return false;
	s.pos += n + 1;
	// quality: exactly as many raw bytes as the sequence line had, then on to the next newline
	if(s.ensure(r.seq_len) < r.seq_len) return false;
	r.qual = s.pos;
	s.pos += r.seq_len;
	s.ensure(1);
	n = s.line_len();
	r.got = true;
	if(n == SIZE_MAX) { s.pos = s.end; return true; }                   // last record without a final newline
	s.pos += n + 1;
	return true;
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
void pack_fastq(const kmahip_ingest *in, const std::vector<Span> *spans, size_t a, size_t b, Part &P) {
	P.clear();
	const kmahip_trim &T = in->trim;
	const double *prob = g_prob.p - in->phred;          // indexed by the raw quality byte
	const int mates = in->paired ? 2 : 1;
	std::vector<uint8_t> *codes = P.codes;
	const bool plain = !T.min_q && !T.hardmask_q;       // only the ends are trimmed, by quality alone (runinput.c:144-171)
	for(size_t i = a; i < b; ++i) {
		int st[2] = {0, 0}, en[2] = {0, 0}, len[2] = {0, 0};
		if(plain) {
			const int minPhred = in->phred + T.min_phred;
			for(int m = 0; m < mates; ++m) {
				const Span &r = spans[m][i];
				const uint8_t *q = in->s[m].buf.data() + r.qual;
				const int L = r.got ? (int) r.seq_len : 0;
				if(T.max_len < L) continue;
				int s0 = 0, e0 = L;
				while(s0 < e0 && q[s0] < minPhred) ++s0;
				while(s0 < e0 && q[e0 - 1] < minPhred) --e0;
				st[m] = s0; en[m] = e0; len[m] = e0 - s0;
			}
			const bool ok0 = T.min_len <= len[0], ok1 = in->paired && T.min_len <= len[1];
			auto put = [&](int m, uint8_t pair) {
				const Span &r = spans[m][i];
				const uint8_t *base = in->s[m].buf.data();
				append_raw(P, base + r.seq + st[m], en[m] - st[m], (const char *) base + r.name, r.got ? r.name_len : 0, pair);
			};
			if(ok0 && ok1) { put(0, 1); put(1, 2); }
			else if(ok0) put(0, 0);
			else if(ok1) put(1, 0);
			else continue;
			++P.records;
			continue;
		}
		for(int m = 0; m < mates; ++m) {
			const Span &r = spans[m][i];
			const uint8_t *base = in->s[m].buf.data();
			const int L = r.got ? (int) r.seq_len : 0;       // a mate file that ran out yields empty mates (the `|` at :516)
			codes[m].resize((size_t) L);
			for(int x = 0; x < L; ++x) codes[m][(size_t) x] = g_trans.t[base[r.seq + (size_t) x]];
			len[m] = phred_stat(codes[m].data(), base + r.qual, L, prob, in->phred + T.min_phred, T.min_q, T.hardmask_q, T.min_len,
			                    T.max_len, &st[m], &en[m]);
		}
		const bool ok0 = T.min_len <= len[0], ok1 = in->paired && T.min_len <= len[1];
		auto put = [&](int m, uint8_t pair) {
			const Span &r = spans[m][i];
			append_read(P, codes[m].data() + st[m], en[m] - st[m], (const char *) in->s[m].buf.data() + r.name, r.got ? r.name_len : 0, pair);
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
	in->seq.insert(in->seq.end(), P.seq.begin(), P.seq.end());
	in->len.insert(in->len.end(), P.len.begin(), P.len.end());
	in->N.insert(in->N.end(), P.N.begin(), P.N.end());
	in->names.insert(in->names.end(), P.names.begin(), P.names.end());
	in->pair.insert(in->pair.end(), P.pair.begin(), P.pair.end());
	for(size_t i = 0; i < P.len.size(); ++i) {
		in->seq_off.push_back(in->seq_off[r0 + i] + ((P.len[i] + 31) >> 5) + 1);
		in->N_off.push_back(in->N_off[r0 + i] + P.nN[i]);
		in->name_off.push_back(in->name_off[r0 + i] + P.name_len[i]);
	}
}

}  // namespace

extern "C" void kmahip_trim_default(kmahip_trim *t) {
	if(!t) return;
	t->min_phred = 20; t->min_q = 0; t->hardmask_q = 0; t->min_len = 16; t->max_len = INT_MAX;
}

extern "C" int kmahip_ingest_open(const char *path1, const char *path2, const kmahip_trim *trim, kmahip_ingest **out) {
	if(!path1 || !out) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	kmahip_ingest *in = new kmahip_ingest();
	if(trim) in->trim = *trim; else kmahip_trim_default(&in->trim);
	// kma.c:1555-1557, runinput.c:380-382
	if(in->trim.min_phred < in->trim.hardmask_q) in->trim.min_phred = in->trim.hardmask_q;
	if(in->trim.min_phred < in->trim.min_q) in->trim.min_phred = in->trim.min_q;
	in->paired = path2 != nullptr;
	{	// worker threads for trimming + packing: KMAHIP_INGEST_THREADS, else the hardware threads, at most 16
		const char *e = getenv("KMAHIP_INGEST_THREADS");
		const int hw = (int) std::thread::hardware_concurrency();
		in->threads = e ? atoi(e) : std::min(16, hw > 0 ? hw : 1);
		if(in->threads < 1) in->threads = 1;
	}
	const char *paths[2] = {path1, path2};
	int kind[2] = {0, 0};
	for(int i = 0; i < (in->paired ? 2 : 1); ++i) {
		if(!in->s[i].open(paths[i])) { kmahip_set_error("cannot open %s", paths[i]); kmahip_ingest_close(in); return KMAHIP_EIO; }
		const size_t have = in->s[i].ensure(FIRST_CHUNK);
		const uint8_t c = have ? *in->s[i].at() : 0;
		kind[i] = c == '@' ? 1 : (c == '>' ? 2 : 0);
		if(have && !kind[i]) { kmahip_set_error("cannot determine format of file %s", paths[i]); kmahip_ingest_close(in); return KMAHIP_EFORMAT; }
	}
	if(in->paired && kind[0] != kind[1]) { kmahip_set_error("%s and %s are in different formats", path1, path2); kmahip_ingest_close(in); return KMAHIP_EFORMAT; }
	in->fastq = kind[0] != 2;
	if(kind[0] == 1) {
		in->phred = guess_phred(in->s[0].at(), std::min(in->s[0].end - in->s[0].pos, FIRST_CHUNK));
		if(in->paired && in->phred == 0) in->phred = guess_phred(in->s[1].at(), std::min(in->s[1].end - in->s[1].pos, FIRST_CHUNK));
	}
	*out = in;
	return KMAHIP_OK;
}

extern "C" int kmahip_ingest_next(kmahip_ingest *in, int64_t max_records, kmahip_read_batch *batch) {
	if(!in || !batch || max_records < 0) { kmahip_set_error("bad argument"); return KMAHIP_EINVAL; }
	in->seq.clear(); in->len.clear(); in->N.clear(); in->names.clear(); in->pair.clear();
	in->seq_off.assign(1, 0); in->N_off.assign(1, 0); in->name_off.assign(1, 0);
	const kmahip_trim &T = in->trim;
	const int mates = in->paired ? 2 : 1;
	int64_t records = 0;
	int max_len = 0;
	if(in->fastq) {
		// Records are located one after the other (memchr over the decompressed bytes), then trimmed and packed by a few
		// threads side by side; a batch takes in `max_records` input records at a time until it holds that many kept ones
		// or the input ends.
		while(records < max_records) {
			const size_t want = (size_t) std::min<int64_t>(max_records - records, 1 << 20);
			std::vector<Span> *spans = in->spans;
			for(int m = 0; m < mates; ++m) {
				spans[m].clear();
				Stream &s = in->s[m];
				if(s.pos) { memmove(s.buf.data(), s.buf.data() + s.pos, s.end - s.pos); s.end -= s.pos; s.pos = 0; }
				s.pinned = true;
				spans[m].reserve(want);
			}
			const bool dbg = getenv("KMAHIP_DEBUG_TIMING") != nullptr;
			const auto t0 = std::chrono::steady_clock::now();
			size_t n = 0;
			for(; n < want; ++n) {
				Span r[2];
				bool any = false;
				for(int m = 0; m < mates; ++m) any |= locate_fq(in->s[m], r[m], &in->malformed);
				if(!any) break;
				for(int m = 0; m < mates; ++m) spans[m].push_back(r[m]);
			}
			for(int m = 0; m < mates; ++m) in->s[m].pinned = false;
			if(n == 0) break;
			in->n_read += (int64_t) n;
			const auto t1 = std::chrono::steady_clock::now();
			const int nt = (int) std::max<size_t>(1, std::min<size_t>((size_t) in->threads, n / 2048));
			if(in->parts.size() < (size_t) nt) in->parts.resize((size_t) nt);
			std::vector<Part> &parts = in->parts;
			std::vector<std::thread> pool;
			for(int t = 1; t < nt; ++t) pool.emplace_back(pack_fastq, in, spans, n * (size_t) t / (size_t) nt, n * (size_t) (t + 1) / (size_t) nt, std::ref(parts[(size_t) t]));
			pack_fastq(in, spans, 0, n / (size_t) nt, parts[0]);
			for(std::thread &th : pool) th.join();
			const auto t2 = std::chrono::steady_clock::now();
			for(int t = 0; t < nt; ++t) { const Part &P = parts[(size_t) t]; append_part(in, P); records += P.records; max_len = std::max(max_len, P.max_len); }
			if(dbg) {
				const auto t3 = std::chrono::steady_clock::now();
				auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
				fprintf(stderr, "[kmahip] ingest: %zu records: read + locate %.1f ms, trim + pack (%d threads) %.1f ms, gather %.1f ms\n", n, ms(t0, t1), nt, ms(t1, t2), ms(t2, t3));
			}
			if(n < want) break;
		}
	} else {
		Rec r[2];
		Part P;
		while(records < max_records) {
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
	batch->names = in->names.data(); batch->name_off = in->name_off.data(); batch->pair = in->pair.data();
	batch->records = records;
	if(in->malformed && !in->reported && records == 0) {
		// like the reference, which prints "Malformed input." and ends with a non-zero exit status after the good records
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
	delete in;
}
