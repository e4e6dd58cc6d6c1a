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

typedef int (*nccl_getid_fn)(void *);
struct NcclId { char b[128]; };                // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128), passed by value
typedef int (*nccl_init_fn2)(void **, int, NcclId, int);
typedef int (*nccl_allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_sendrecv_fn)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_recv_fn)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_destroy_fn)(void *);

struct Rccl {
	void *lib = nullptr;
	nccl_getid_fn get_id = nullptr;
	nccl_init_fn2 init = nullptr;
	nccl_allreduce_fn allreduce = nullptr;
	nccl_sendrecv_fn send = nullptr;
	nccl_recv_fn recv = nullptr;
	nccl_group_fn group_start = nullptr, group_end = nullptr;
	nccl_destroy_fn destroy = nullptr;
	bool load() {
		auto sym = [&](const char *n) -> void * {
			void *p = dlsym(RTLD_DEFAULT, n);
			if(!p && lib) p = dlsym(lib, n);
			return p;
		};
		if(!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
			lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
			if(!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
			if(!lib) return false;
		}
		get_id = (nccl_getid_fn) sym("ncclGetUniqueId"); init = (nccl_init_fn2) sym("ncclCommInitRank");
		allreduce = (nccl_allreduce_fn) sym("ncclAllReduce"); send = (nccl_sendrecv_fn) sym("ncclSend"); recv = (nccl_recv_fn) sym("ncclRecv");
		group_start = (nccl_group_fn) sym("ncclGroupStart"); group_end = (nccl_group_fn) sym("ncclGroupEnd"); destroy = (nccl_destroy_fn) sym("ncclCommDestroy");
		return get_id && init && allreduce && send && recv && group_start && group_end && destroy;
	}
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
	void *nccl_comm = nullptr;
	ShmHeader *hdr() const { return (ShmHeader *) seg; }
	uint8_t *box(int r) const { return seg + 4096 + (size_t) r * MAILBOX; }
};

namespace {

int fail(kmahip_comm *c, const char *what) {
	if(c && c->seg) c->hdr()->failed.store(1);
	kmahip_set_error("communicator (rank %d of %d): %s", c ? c->rank : -1, c ? c->world : 0, what);
	return KMAHIP_EDEVICE;
}

int barrier(kmahip_comm *c) {
	if(c->world == 1) return KMAHIP_OK;
	ShmHeader *h = c->hdr();
	const uint32_t gen = h->generation.load();
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

}  // namespace

extern "C" int kmahip_comm_init(int rank, int world, const char *key, const char *backend, kmahip_comm **out) {
	if(!out || rank < 0 || world < 1 || rank >= world || world > MAX_WORLD || !key || !*key) { kmahip_set_error("kmahip_comm_init: bad rank / world / key"); return KMAHIP_EINVAL; }
	kmahip_comm *c = new kmahip_comm();
	c->rank = rank; c->world = world; c->key = key;
	for(char &ch : c->key) if(!isalnum((unsigned char) ch) && ch != '_' && ch != '-') ch = '_';
	if(const char *t = getenv("KMAHIP_COMM_TIMEOUT")) c->timeout_s = atof(t);
	*out = c;
	if(world == 1) return KMAHIP_OK;
	c->shm_name = "/kmahip_" + c->key;
	c->seg_bytes = 4096 + (size_t) world * MAILBOX;
	// rank 0 creates and initialises the segment, the others wait for it
	int fd = -1;
	const auto t0 = std::chrono::steady_clock::now();
	if(rank == 0) {
		shm_unlink(c->shm_name.c_str());
		fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
		if(fd < 0 || ftruncate(fd, (off_t) c->seg_bytes) != 0) { if(fd >= 0) close(fd); return fail(c, "cannot create the shared-memory segment"); }
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
	if(m == MAP_FAILED) return fail(c, "cannot map the shared-memory segment");
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
	int rc = barrier(c);
	if(rc) return rc;
	if(rank == 0) shm_unlink(c->shm_name.c_str());          // every rank has it mapped: the name can go
	if(backend && !strcmp(backend, "rccl")) {
		if(!c->nccl.load()) return fail(c, "RCCL (librccl.so) not found");
		NcclId id;
		memset(&id, 0, sizeof id);
		if(rank == 0) {
			if(c->nccl.get_id(&id)) return fail(c, "ncclGetUniqueId failed");
			memcpy(c->box(0), &id, sizeof id);
		}
		if((rc = barrier(c))) return rc;
		memcpy(&id, c->box(0), sizeof id);
		if((rc = barrier(c))) return rc;
		if(c->nccl.init(&c->nccl_comm, world, id, rank)) return fail(c, "ncclCommInitRank failed (one device per rank is needed; use the shm backend to rehearse on fewer devices)");
		c->rccl = true;
	} else if(backend && strcmp(backend, "shm")) { kmahip_set_error("kmahip_comm_init: unknown backend %s (rccl or shm)", backend); return KMAHIP_EINVAL; }
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
	if(c->world == 1 || n == 0) return KMAHIP_OK;
	hipStream_t s = (hipStream_t) stream;
	if(c->rccl) {
		const int ncclUint64 = 5, ncclSum = 0;
		if(c->nccl.allreduce(d_buf, d_buf, n, ncclUint64, ncclSum, c->nccl_comm, s)) return fail(c, "ncclAllReduce failed");
		HIP_TRY(hipStreamSynchronize(s));
		return KMAHIP_OK;
	}
	// staged: every rank writes its vector into a payload file, reads and adds the others'
	std::vector<uint64_t> mine(n), sum(n, 0);
	HIP_TRY(hipMemcpyAsync(mine.data(), d_buf, n * 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	const uint64_t seq = c->seq++;
	const std::string name = payload_name(c, seq, c->rank);
	int fd = shm_open(name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
	if(fd < 0 || ftruncate(fd, (off_t) (n * 8)) != 0 || pwrite(fd, mine.data(), n * 8, 0) != (ssize_t) (n * 8)) { if(fd >= 0) close(fd); return fail(c, "cannot write a payload file"); }
	close(fd);
	int rc;
	if((rc = barrier(c))) { shm_unlink(name.c_str()); return rc; }
	std::vector<uint64_t> other(n);
	for(int r = 0; r < c->world; ++r) {
		const uint64_t *src = mine.data();
		if(r != c->rank) {
			fd = shm_open(payload_name(c, seq, r).c_str(), O_RDONLY, 0600);
			if(fd < 0 || pread(fd, other.data(), n * 8, 0) != (ssize_t) (n * 8)) { if(fd >= 0) close(fd); shm_unlink(name.c_str()); return fail(c, "cannot read a payload file"); }
			close(fd);
			src = other.data();
		}
		for(size_t i = 0; i < n; ++i) sum[i] += src[i];
	}
	rc = barrier(c);
	shm_unlink(name.c_str());
	if(rc) return rc;
	HIP_TRY(hipMemcpyAsync(d_buf, sum.data(), n * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipStreamSynchronize(s));
	return KMAHIP_OK;
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
	if(W == 1) {
		if(send_bytes[0] != recv_bytes[0]) { kmahip_set_error("block sizes disagree"); return KMAHIP_EINVAL; }
		if(send_bytes[0]) {
			if(device) { HIP_TRY(hipMemcpyAsync(recv, send, (size_t) send_bytes[0], hipMemcpyDeviceToDevice, s)); HIP_TRY(hipStreamSynchronize(s)); }
			else memcpy(recv, send, (size_t) send_bytes[0]);
		}
		return KMAHIP_OK;
	}
	if(device && c->rccl) {
		const int ncclUint8 = 1;
		if(c->nccl.group_start()) return fail(c, "ncclGroupStart failed");
		for(int r = 0; r < W; ++r) {
			if(send_bytes[r] && c->nccl.send((const uint8_t *) send + so[(size_t) r], (size_t) send_bytes[r], ncclUint8, r, c->nccl_comm, s)) return fail(c, "ncclSend failed");
			if(recv_bytes[r] && c->nccl.recv((uint8_t *) recv + ro[(size_t) r], (size_t) recv_bytes[r], ncclUint8, r, c->nccl_comm, s)) return fail(c, "ncclRecv failed");
		}
		if(c->nccl.group_end()) return fail(c, "ncclGroupEnd failed");
		HIP_TRY(hipStreamSynchronize(s));
		return KMAHIP_OK;
	}
	// through a payload file per source: [W + 1 offsets][blocks]
	const uint64_t seq = c->seq++;
	const std::string name = payload_name(c, seq, c->rank);
	const size_t head = ((size_t) W + 1) * 8, total = head + (size_t) so[(size_t) W];
	int fd = shm_open(name.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0600);
	if(fd < 0 || ftruncate(fd, (off_t) total) != 0) { if(fd >= 0) close(fd); return fail(c, "cannot create a payload file"); }
	uint8_t *m = (uint8_t *) mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if(m == MAP_FAILED) { shm_unlink(name.c_str()); return fail(c, "cannot map a payload file"); }
	memcpy(m, so.data(), head);
	if(so[(size_t) W]) {
		if(device) { HIP_TRY(hipMemcpyAsync(m + head, send, (size_t) so[(size_t) W], hipMemcpyDeviceToHost, s)); HIP_TRY(hipStreamSynchronize(s)); }
		else memcpy(m + head, send, (size_t) so[(size_t) W]);
	}
	int rc;
	if((rc = barrier(c))) { munmap(m, total); shm_unlink(name.c_str()); return rc; }
	std::vector<uint8_t> stage;
	uint8_t *dst = (uint8_t *) recv;
	if(device) { stage.resize((size_t) ro[(size_t) W]); dst = stage.data(); }
	for(int r = 0; r < W && !rc; ++r) {
		if(!recv_bytes[r]) continue;
		if(r == c->rank) { memcpy(dst + ro[(size_t) r], m + head + so[(size_t) r], (size_t) recv_bytes[r]); continue; }
		fd = shm_open(payload_name(c, seq, r).c_str(), O_RDONLY, 0600);
		int64_t off[2] = {0, 0};
		if(fd < 0 || pread(fd, off, 16, (off_t) c->rank * 8) != 16 || off[1] - off[0] != recv_bytes[r] ||
		   pread(fd, dst + ro[(size_t) r], (size_t) recv_bytes[r], (off_t) (head + (size_t) off[0])) != (ssize_t) recv_bytes[r]) rc = fail(c, "payload of another rank missing or of another size than agreed");
		if(fd >= 0) close(fd);
	}
	const int rc2 = barrier(c);
	munmap(m, total);
	shm_unlink(name.c_str());
	if(rc || rc2) return rc ? rc : rc2;
	if(device && ro[(size_t) W]) { HIP_TRY(hipMemcpyAsync(recv, stage.data(), (size_t) ro[(size_t) W], hipMemcpyHostToDevice, s)); HIP_TRY(hipStreamSynchronize(s)); }
	return KMAHIP_OK;
}
