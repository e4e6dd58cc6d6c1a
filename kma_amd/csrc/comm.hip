// comm.hip -- what crosses the ranks of a read-sharded run (one process per GPU, all on one node): a small communicator behind the
// C-ABI, so that a C host program needs neither MPI nor torch.distributed.
//
// The reference has nothing to restate here (one process, threads over pipes, kmapipe.c:55-146); what is exchanged is listed in
// the SUM of the two ConClave score vectors before runConClave (runkma.c:563-594), the SUM of ConClave's
// per-template outputs before the `.res` statistics (runkma.c:608-613, 770-783), and the traced reads travelling to the rank that
// owns their template (assembly order, conclave.c:164-196).
//
//   bootstrap + small host data   a POSIX shared-memory segment named after the rendezvous key (/dev/shm/kmahip_<key>): a
//                                 sense-reversing barrier and one mailbox per rank. Always there: the ranks are processes of one node.
//   device vectors and payloads   backend "rccl": ncclAllReduce / grouped ncclSend + ncclRecv over xGMI, the ncclUniqueId handed
//                                 round through the mailbox (librccl is resolved at run time, like kmahip_allreduce_scores);
//                                 backend "shm": staged through host memory and shared-memory files -- for boxes with fewer
//                                 devices than ranks (RCCL refuses two ranks on one device) and for the tests.
#include "kmahip_internal.h"
#include <atomic>
#include <chrono>
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {

constexpr size_t MAILBOX = 64u << 10;          // bytes a rank can post at a time
constexpr int MAX_WORLD = 64;

struct ShmHeader {
	std::atomic<uint32_t> magic;               // set by rank 0 when the segment is initialised
	std::atomic<uint32_t> arrived, generation;
	std::atomic<uint32_t> failed;              // a rank gave up: the others stop waiting
	uint32_t world;
};

// The entry points are resolved at run time (the library does not link against librccl), but their TYPES, the size of
// ncclUniqueId and the datatype / reduction enums come from rccl.h at build time: a header that changes them breaks the build,
// not a run on eight devices.
static_assert(sizeof(ncclUniqueId) == NCCL_UNIQUE_ID_BYTES && NCCL_UNIQUE_ID_BYTES <= MAILBOX, "ncclUniqueId travels through a mailbox");
static_assert(sizeof(uint64_t) == 8 && ncclUint64 != ncclInt64 && ncclUint8 != ncclInt8, "rccl.h datatypes");

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclGetVersion) get_version = nullptr;
	decltype(&ncclGetUniqueId) get_id = nullptr;
	decltype(&ncclCommInitRank) init = nullptr;
	decltype(&ncclCommCount) count = nullptr;
	decltype(&ncclCommUserRank) user_rank = nullptr;
	decltype(&ncclAllReduce) allreduce = nullptr;
	decltype(&ncclSend) send = nullptr;
	decltype(&ncclRecv) recv = nullptr;
	decltype(&ncclGroupStart) group_start = nullptr;
	decltype(&ncclGroupEnd) group_end = nullptr;
	decltype(&ncclCommDestroy) destroy = nullptr;
	decltype(&ncclGetErrorString) error_string = nullptr;
	bool load() {
		if(!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
			lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
			if(!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
			if(!lib) return false;
		}
		auto sym = [&](const char *n) -> void * {
			void *p = dlsym(RTLD_DEFAULT, n);
			if(!p && lib) p = dlsym(lib, n);
			return p;
		};
		get_version = (decltype(get_version)) sym("ncclGetVersion");
		get_id = (decltype(get_id)) sym("ncclGetUniqueId"); init = (decltype(init)) sym("ncclCommInitRank");
		count = (decltype(count)) sym("ncclCommCount"); user_rank = (decltype(user_rank)) sym("ncclCommUserRank");
		allreduce = (decltype(allreduce)) sym("ncclAllReduce"); send = (decltype(send)) sym("ncclSend"); recv = (decltype(recv)) sym("ncclRecv");
		group_start = (decltype(group_start)) sym("ncclGroupStart"); group_end = (decltype(group_end)) sym("ncclGroupEnd");
		destroy = (decltype(destroy)) sym("ncclCommDestroy"); error_string = (decltype(error_string)) sym("ncclGetErrorString");
		return get_version && get_id && init && count && user_rank && allreduce && send && recv && group_start && group_end && destroy;
	}
	const char *why(ncclResult_t r) const { return error_string ? error_string(r) : "?"; }
};

}  // namespace

struct kmahip_comm {
	int rank = 0, world = 1;
	bool rccl = false;
	std::string key, shm_name;
	uint8_t *seg = nullptr;
	size_t seg_bytes = 0;
	uint32_t sense = 0;
	uint64_t seq = 0;                          // exchanges so far (names the payload files of the shm backend)
	double timeout_s = 600;
	Rccl nccl;
	ncclComm_t nccl_comm = nullptr;
	int nccl_version = 0, nccl_nranks = 0, nccl_rank = -1;    // what RCCL itself reports of the communicator
	uint64_t n_allreduce = 0, n_alltoallv = 0;                 // exchanges that went through RCCL
	ShmHeader *hdr() const { return (ShmHeader *) seg; }
	uint8_t *box(int r) const { return seg + 4096 + (size_t) r * MAILBOX; }
};

namespace {

int fail(kmahip_comm *c, const char *what) {
	if(c && c->seg) c->hdr()->failed.store(1);
	kmahip_set_error("communicator (rank %d of %d): %s", c ? c->rank : -1, c ? c->world : 0, what);
	return KMAHIP_EDEVICE;
}
int fail_nccl(kmahip_comm *c, const char *what, ncclResult_t r) {
	if(c && c->seg) c->hdr()->failed.store(1);
	kmahip_set_error("communicator (rank %d of %d): %s: %s (%d)", c->rank, c->world, what, c->nccl.why(r), (int) r);
	return KMAHIP_EDEVICE;
}
// a HIP call inside an exchange: a rank that leaves tells the others (they would wait for it until the timeout)
#define COMM_HIP(c, call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { (void) hipGetLastError(); \
	kmahip_set_error("communicator (rank %d of %d): %s: %s", (c)->rank, (c)->world, #call, hipGetErrorString(e_)); \
	if((c)->seg) (c)->hdr()->failed.store(1); rc = KMAHIP_EDEVICE; } } while(0)

int barrier(kmahip_comm *c) {
	if(c->world == 1) return KMAHIP_OK;
	ShmHeader *h = c->hdr();
	const uint32_t gen = h->generation.load();
	if(h->failed.load()) return fail(c, "another rank failed");
	if(h->arrived.fetch_add(1) + 1 == (uint32_t) c->world) {
		h->arrived.store(0);
		h->generation.store(gen + 1);
		return KMAHIP_OK;
	}
	const auto t0 = std::chrono::steady_clock::now();
	for(unsigned spin = 0; h->generation.load() == gen; ++spin) {
		if(h->failed.load()) return fail(c, "another rank failed");
		if(spin < 2000) std::this_thread::yield();
		else {
			std::this_thread::sleep_for(std::chrono::microseconds(50));
			if((spin & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) return fail(c, "timed out waiting for the other ranks");
		}
	}
	return KMAHIP_OK;
}

// a payload file of the shm backend: what rank `src` sends in exchange number `seq`
std::string payload_name(const kmahip_comm *c, uint64_t seq, int src) {
	return "/kmahip_" + c->key + "_x" + std::to_string((unsigned long long) seq) + "_" + std::to_string(src);
}

// one read() / write() moves at most 0x7ffff000 bytes on Linux: a block of 2 GiB and more (the kept reads of a sharded `-Mt1` run
// all travel to rank 0) takes several
bool pread_all(int fd, void *buf, size_t n, off_t off) {
	uint8_t *p = (uint8_t *) buf;
	while(n) {
		const ssize_t k = pread(fd, p, n < ((size_t) 1 << 30) ? n : ((size_t) 1 << 30), off);
		if(k < 0 && errno == EINTR) continue;
		if(k <= 0) return false;
		p += k; off += k; n -= (size_t) k;
	}
	return true;
}
bool pwrite_all(int fd, const void *buf, size_t n, off_t off) {
	const uint8_t *p = (const uint8_t *) buf;
	while(n) {
		const ssize_t k = pwrite(fd, p, n < ((size_t) 1 << 30) ? n : ((size_t) 1 << 30), off);
		if(k < 0 && errno == EINTR) continue;
		if(k <= 0) return false;
		p += k; off += k; n -= (size_t) k;
	}
	return true;
}

// RCCL's own view of the communicator just made, and one exchange of each kind through it: every rank adds rank + 1 into one u64
// (the sum is world (world + 1) / 2) and sends its rank number to every rank, itself included, in one group. A build whose
// datatype enums, unique-id layout or group semantics differ from the rccl.h this file was compiled with fails HERE, at start-up.
int rccl_self_test(kmahip_comm *c) {
	ncclResult_t r;
	if((r = c->nccl.get_version(&c->nccl_version)) != ncclSuccess) return fail_nccl(c, "ncclGetVersion", r);
	if((r = c->nccl.count(c->nccl_comm, &c->nccl_nranks)) != ncclSuccess) return fail_nccl(c, "ncclCommCount", r);
	if((r = c->nccl.user_rank(c->nccl_comm, &c->nccl_rank)) != ncclSuccess) return fail_nccl(c, "ncclCommUserRank", r);
	if(c->nccl_nranks != c->world || c->nccl_rank != c->rank) return fail(c, "RCCL reports another rank / size than the communicator was made with");
	const int W = c->world;
	uint64_t *d = nullptr;
	int rc = KMAHIP_OK;
	COMM_HIP(c, hipMalloc((void **) &d, (size_t) (2 * W + 1) * 8));
	if(rc) return rc;
	std::vector<uint64_t> h((size_t) 2 * W + 1, 0);
	h[0] = (uint64_t) c->rank + 1 + ((uint64_t) (c->rank + 1) << 40);      // both halves of the 64-bit lane
	for(int p = 0; p < W; ++p) h[(size_t) 1 + p] = (uint64_t) c->rank * 1000 + p;
	COMM_HIP(c, hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
	if(!rc && (r = c->nccl.allreduce(d, d, 1, ncclUint64, ncclSum, c->nccl_comm, nullptr)) != ncclSuccess) rc = fail_nccl(c, "ncclAllReduce (self-test)", r);
	if(!rc && (r = c->nccl.group_start()) != ncclSuccess) rc = fail_nccl(c, "ncclGroupStart (self-test)", r);
	for(int p = 0; p < W && !rc; ++p) {
		if((r = c->nccl.send(d + 1 + p, 8, ncclUint8, p, c->nccl_comm, nullptr)) != ncclSuccess) rc = fail_nccl(c, "ncclSend (self-test)", r);
		else if((r = c->nccl.recv(d + 1 + W + p, 8, ncclUint8, p, c->nccl_comm, nullptr)) != ncclSuccess) rc = fail_nccl(c, "ncclRecv (self-test)", r);
	}
	if(!rc && (r = c->nccl.group_end()) != ncclSuccess) rc = fail_nccl(c, "ncclGroupEnd (self-test)", r);
	if(!rc) COMM_HIP(c, hipStreamSynchronize(nullptr));
	if(!rc) COMM_HIP(c, hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
	(void) hipFree(d);
	if(rc) return rc;
	const uint64_t tri = (uint64_t) W * (W + 1) / 2;
	if(h[0] != tri + (tri << 40)) return fail(c, "RCCL self-test: the u64 SUM all-reduce gave a wrong sum (datatype enum?)");
	for(int p = 0; p < W; ++p) if(h[(size_t) 1 + W + p] != (uint64_t) p * 1000 + c->rank) return fail(c, "RCCL self-test: grouped send / recv delivered the wrong block");
	return KMAHIP_OK;
}

}  // namespace

extern "C" int kmahip_comm_init(int rank, int world, const char *key, const char *backend, kmahip_comm **out) {
	if(!out || rank < 0 || world < 1 || rank >= world || world > MAX_WORLD || !key || !*key) { kmahip_set_error("kmahip_comm_init: bad rank / world / key"); return KMAHIP_EINVAL; }
	if(backend && strcmp(backend, "rccl") && strcmp(backend, "shm")) { kmahip_set_error("kmahip_comm_init: unknown backend %s (rccl or shm)", backend); return KMAHIP_EINVAL; }
	kmahip_comm *c = new kmahip_comm();
	c->rank = rank; c->world = world; c->key = key;
	for(char &ch : c->key) if(!isalnum((unsigned char) ch) && ch != '_' && ch != '-') ch = '_';
	if(const char *t = getenv("KMAHIP_COMM_TIMEOUT")) c->timeout_s = atof(t);
	*out = c;
	const bool want_rccl = backend && !strcmp(backend, "rccl");
	// One rank needs no transport. KMAHIP_COMM_FORCE_RCCL=1 makes a real one-rank RCCL communicator all the same and sends every
	// exchange through it: that is how the transport is exercised on a box with a single device (tests/test_comm_rccl_gpu.py).
	const char *force = getenv("KMAHIP_COMM_FORCE_RCCL");
	if(world == 1 && !(want_rccl && force && *force == '1')) return KMAHIP_OK;
	int rc;
	if(world > 1) {
		c->shm_name = "/kmahip_" + c->key;
		c->seg_bytes = 4096 + (size_t) world * MAILBOX;
		// rank 0 creates and initialises the segment, the others wait for it
		int fd = -1;
		const auto t0 = std::chrono::steady_clock::now();
		if(rank == 0) {
			shm_unlink(c->shm_name.c_str());
			fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
			if(fd < 0 || ftruncate(fd, (off_t) c->seg_bytes) != 0) { if(fd >= 0) { close(fd); shm_unlink(c->shm_name.c_str()); } return fail(c, "cannot create the shared-memory segment"); }
		} else {
			for(;;) {
				fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
				struct stat sb;
				if(fd >= 0 && fstat(fd, &sb) == 0 && (size_t) sb.st_size >= c->seg_bytes) break;
				if(fd >= 0) close(fd);
				fd = -1;
				if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) return fail(c, "rank 0 never created the shared-memory segment");
				std::this_thread::sleep_for(std::chrono::milliseconds(2));
			}
		}
		void *m = mmap(nullptr, c->seg_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if(m == MAP_FAILED) { if(rank == 0) shm_unlink(c->shm_name.c_str()); return fail(c, "cannot map the shared-memory segment"); }
		c->seg = (uint8_t *) m;
		ShmHeader *h = c->hdr();
		if(rank == 0) {
			h->arrived.store(0); h->generation.store(0); h->failed.store(0); h->world = (uint32_t) world;
			h->magic.store(0x4b4d4148u);
		} else {
			while(h->magic.load() != 0x4b4d4148u) {
				if(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) return fail(c, "the shared-memory segment was never initialised");
				std::this_thread::sleep_for(std::chrono::milliseconds(1));
			}
			if(h->world != (uint32_t) world) return fail(c, "ranks disagree on the world size (a stale segment of another run?)");
		}
		rc = barrier(c);
		if(rank == 0) shm_unlink(c->shm_name.c_str());          // every rank has it mapped (or has given up): the name can go
		if(rc) return rc;
	}
	if(want_rccl) {
		if(!c->nccl.load()) return fail(c, "RCCL (librccl.so) not found");
		ncclUniqueId id;
		memset(&id, 0, sizeof id);
		ncclResult_t r = ncclSuccess;
		if(rank == 0 && (r = c->nccl.get_id(&id)) != ncclSuccess) return fail_nccl(c, "ncclGetUniqueId", r);
		if(world > 1) {
			if(rank == 0) memcpy(c->box(0), &id, sizeof id);
			if((rc = barrier(c))) return rc;
			memcpy(&id, c->box(0), sizeof id);
			if((rc = barrier(c))) return rc;
		}
		if((r = c->nccl.init(&c->nccl_comm, world, id, rank)) != ncclSuccess) return fail_nccl(c, "ncclCommInitRank (one device per rank is needed; use the shm backend to rehearse on fewer devices)", r);
		c->rccl = true;
		if((rc = rccl_self_test(c))) return rc;
		if(const char *v = getenv("KMAHIP_COMM_VERBOSE")) if(*v == '1') {
			char line[256];
			kmahip_comm_describe(c, line, sizeof line);
			fprintf(stderr, "kmahip_comm: %s\n", line);
		}
	}
	return KMAHIP_OK;
}

extern "C" void kmahip_comm_destroy(kmahip_comm *c) {
	if(!c) return;
	if(c->nccl_comm && c->nccl.destroy) c->nccl.destroy(c->nccl_comm);
	if(c->seg) munmap(c->seg, c->seg_bytes);
	delete c;
}

extern "C" int kmahip_comm_rank(const kmahip_comm *c) { return c ? c->rank : 0; }
extern "C" int kmahip_comm_world(const kmahip_comm *c) { return c ? c->world : 1; }
extern "C" int kmahip_comm_is_rccl(const kmahip_comm *c) { return c && c->rccl; }
extern "C" int kmahip_comm_barrier(kmahip_comm *c) { return c ? barrier(c) : KMAHIP_OK; }

// one line for a log or a JSON record: what carries the device data and what the transport itself reports
extern "C" int kmahip_comm_describe(const kmahip_comm *c, char *buf, size_t cap) {
	if(!buf || !cap) return 0;
	if(!c) return snprintf(buf, cap, "backend=none rank=0 world=1");
	if(c->rccl) return snprintf(buf, cap, "backend=rccl rank=%d world=%d rccl_nranks=%d rccl_rank=%d rccl_version=%d allreduces=%llu alltoallvs=%llu",
	                            c->rank, c->world, c->nccl_nranks, c->nccl_rank, c->nccl_version, (unsigned long long) c->n_allreduce, (unsigned long long) c->n_alltoallv);
	return snprintf(buf, cap, "backend=%s rank=%d world=%d", c->world == 1 ? "none" : "shm", c->rank, c->world);
}

// every rank posts `bytes` (<= 64 KiB) and gets all of them back in rank order (HOST memory)
extern "C" int kmahip_comm_allgather(kmahip_comm *c, const void *mine, size_t bytes, void *all) {
	if(!c || (!mine && bytes) || !all) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(bytes > MAILBOX) { kmahip_set_error("kmahip_comm_allgather: %zu bytes per rank, the mailbox takes %zu", bytes, MAILBOX); return KMAHIP_EINVAL; }
	if(c->world == 1) { memcpy(all, mine, bytes); return KMAHIP_OK; }
	int rc;
	memcpy(c->box(c->rank), mine, bytes);
	if((rc = barrier(c))) return rc;
	for(int r = 0; r < c->world; ++r) memcpy((uint8_t *) all + (size_t) r * bytes, c->box(r), bytes);
	return barrier(c);
}

// in-place SUM over the ranks of n u64 values in DEVICE memory (exact and order-free: the ConClave vectors)
extern "C" int kmahip_comm_allreduce_u64(kmahip_comm *c, uint64_t *d_buf, size_t n, void *stream) {
	if(!c || (!d_buf && n)) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	if(n == 0 || (c->world == 1 && !c->rccl)) return KMAHIP_OK;
	hipStream_t s = (hipStream_t) stream;
	int rc = KMAHIP_OK;
	if(c->rccl) {
		const ncclResult_t r = c->nccl.allreduce(d_buf, d_buf, n, ncclUint64, ncclSum, c->nccl_comm, s);
		if(r != ncclSuccess) return fail_nccl(c, "ncclAllReduce", r);
		COMM_HIP(c, hipStreamSynchronize(s));
		++c->n_allreduce;
		return rc;
	}
	// staged: every rank writes its vector into a payload file, reads and adds the others'
	std::vector<uint64_t> mine(n), sum(n, 0);
	COMM_HIP(c, hipMemcpyAsync(mine.data(), d_buf, n * 8, hipMemcpyDeviceToHost, s));
	if(!rc) COMM_HIP(c, hipStreamSynchronize(s));
	if(rc) return rc;
	const uint64_t seq = c->seq++;
	const std::string name = payload_name(c, seq, c->rank);
	int fd = shm_open(name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
	if(fd < 0 || ftruncate(fd, (off_t) (n * 8)) != 0 || !pwrite_all(fd, mine.data(), n * 8, 0)) { if(fd >= 0) close(fd); shm_unlink(name.c_str()); return fail(c, "cannot write a payload file"); }
	close(fd);
	if((rc = barrier(c))) { shm_unlink(name.c_str()); return rc; }
	std::vector<uint64_t> other(n);
	for(int r = 0; r < c->world && !rc; ++r) {
		const uint64_t *src = mine.data();
		if(r != c->rank) {
			fd = shm_open(payload_name(c, seq, r).c_str(), O_RDONLY, 0600);
			if(fd < 0 || !pread_all(fd, other.data(), n * 8, 0)) rc = fail(c, "cannot read a payload file");
			if(fd >= 0) close(fd);
			src = other.data();
		}
		if(!rc) for(size_t i = 0; i < n; ++i) sum[i] += src[i];
	}
	const int rc2 = rc ? rc : barrier(c);
	shm_unlink(name.c_str());
	if(rc2) return rc2;
	COMM_HIP(c, hipMemcpyAsync(d_buf, sum.data(), n * 8, hipMemcpyHostToDevice, s));
	if(!rc) COMM_HIP(c, hipStreamSynchronize(s));
	return rc;
}

// all-to-all of byte blocks. send: world blocks back to back, send_bytes[d] of them for rank d; recv_bytes[s] (what rank s sends
// here, agreed on beforehand e.g. through kmahip_comm_allgather) land back to back in source-rank order. device != 0: both
// buffers are DEVICE memory (RCCL: ncclSend / ncclRecv in one group; shm: staged), else HOST memory (always through shm files).
extern "C" int kmahip_comm_alltoallv(kmahip_comm *c, const void *send, const int64_t *send_bytes, void *recv, const int64_t *recv_bytes,
                                     int device, void *stream) {
	if(!c || !send_bytes || !recv_bytes) { kmahip_set_error("null argument"); return KMAHIP_EINVAL; }
	const int W = c->world;
	hipStream_t s = (hipStream_t) stream;
	std::vector<int64_t> so((size_t) W + 1, 0), ro((size_t) W + 1, 0);
	for(int r = 0; r < W; ++r) {
		if(send_bytes[r] < 0 || recv_bytes[r] < 0) { kmahip_set_error("negative block size"); return KMAHIP_EINVAL; }
		so[(size_t) r + 1] = so[(size_t) r] + send_bytes[r]; ro[(size_t) r + 1] = ro[(size_t) r] + recv_bytes[r];
	}
	if((so[(size_t) W] && !send) || (ro[(size_t) W] && !recv)) { kmahip_set_error("null buffer"); return KMAHIP_EINVAL; }
	int rc = KMAHIP_OK;
	if(device && c->rccl) {
		if(send_bytes[c->rank] != recv_bytes[c->rank]) { kmahip_set_error("block sizes disagree"); return KMAHIP_EINVAL; }
		ncclResult_t r;
		if((r = c->nccl.group_start()) != ncclSuccess) return fail_nccl(c, "ncclGroupStart", r);
		for(int p = 0; p < W; ++p) {
			if(send_bytes[p] && (r = c->nccl.send((const uint8_t *) send + so[(size_t) p], (size_t) send_bytes[p], ncclUint8, p, c->nccl_comm, s)) != ncclSuccess) { c->nccl.group_end(); return fail_nccl(c, "ncclSend", r); }
			if(recv_bytes[p] && (r = c->nccl.recv((uint8_t *) recv + ro[(size_t) p], (size_t) recv_bytes[p], ncclUint8, p, c->nccl_comm, s)) != ncclSuccess) { c->nccl.group_end(); return fail_nccl(c, "ncclRecv", r); }
		}
		if((r = c->nccl.group_end()) != ncclSuccess) return fail_nccl(c, "ncclGroupEnd", r);
		COMM_HIP(c, hipStreamSynchronize(s));
		++c->n_alltoallv;
		return rc;
	}
	if(W == 1) {
		if(send_bytes[0] != recv_bytes[0]) { kmahip_set_error("block sizes disagree"); return KMAHIP_EINVAL; }
		if(send_bytes[0]) {
			if(device) { HIP_TRY(hipMemcpyAsync(recv, send, (size_t) send_bytes[0], hipMemcpyDeviceToDevice, s)); HIP_TRY(hipStreamSynchronize(s)); }
			else memcpy(recv, send, (size_t) send_bytes[0]);
		}
		return KMAHIP_OK;
	}
	// through a payload file per source: [W + 1 offsets][blocks]
	const uint64_t seq = c->seq++;
	const std::string name = payload_name(c, seq, c->rank);
	const size_t head = ((size_t) W + 1) * 8, total = head + (size_t) so[(size_t) W];
	int fd = shm_open(name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
	if(fd < 0 || ftruncate(fd, (off_t) total) != 0) { if(fd >= 0) close(fd); shm_unlink(name.c_str()); return fail(c, "cannot create a payload file"); }
	uint8_t *m = (uint8_t *) mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if(m == MAP_FAILED) { shm_unlink(name.c_str()); return fail(c, "cannot map a payload file"); }
	memcpy(m, so.data(), head);
	if(so[(size_t) W]) {
		if(device) { COMM_HIP(c, hipMemcpyAsync(m + head, send, (size_t) so[(size_t) W], hipMemcpyDeviceToHost, s)); if(!rc) COMM_HIP(c, hipStreamSynchronize(s)); }
		else memcpy(m + head, send, (size_t) so[(size_t) W]);
	}
	if(rc || (rc = barrier(c))) { munmap(m, total); shm_unlink(name.c_str()); return rc; }
	std::vector<uint8_t> stage;
	uint8_t *dst = (uint8_t *) recv;
	if(device) { stage.resize((size_t) ro[(size_t) W]); dst = stage.data(); }
	for(int r = 0; r < W && !rc; ++r) {
		if(!recv_bytes[r]) continue;
		if(r == c->rank) { memcpy(dst + ro[(size_t) r], m + head + so[(size_t) r], (size_t) recv_bytes[r]); continue; }
		fd = shm_open(payload_name(c, seq, r).c_str(), O_RDONLY, 0600);
		int64_t off[2] = {0, 0};
		if(fd < 0 || !pread_all(fd, off, 16, (off_t) c->rank * 8) || off[1] - off[0] != recv_bytes[r] ||
		   !pread_all(fd, dst + ro[(size_t) r], (size_t) recv_bytes[r], (off_t) (head + (size_t) off[0]))) rc = fail(c, "payload of another rank missing or of another size than agreed");
		if(fd >= 0) close(fd);
	}
	const int rc2 = rc ? rc : barrier(c);
	munmap(m, total);
	shm_unlink(name.c_str());
	if(rc2) return rc2;
	if(device && ro[(size_t) W]) { COMM_HIP(c, hipMemcpyAsync(recv, stage.data(), (size_t) ro[(size_t) W], hipMemcpyHostToDevice, s)); if(!rc) COMM_HIP(c, hipStreamSynchronize(s)); }
	return rc;
}
