// rk_comm_*: the collectives of the hash-sharded search (include/rubiks_hip.h, "hash-sharded mode") over RCCL, for callers
// of the C ABI that have no torch.distributed -- SURVEY.md 8(b) sketched this layer, INTEGRATION.md route B uses it.
// No counterpart in the reference (single process).  One communicator per process = per GPU; the transfers run on the
// caller's HIP stream between the engine's kernels, on the engine's own fixed-size device buffers:
//   all-gather   (8 + N) doubles per rank                                     -> ncclAllGather
//   all-to-all   one block per peer, equal sizes (the counts travel inside)   -> grouped ncclSend / ncclRecv: a different
//                block goes to every peer, so all seven xGMI links of a GPU carry traffic at once
//   broadcast    three integers per hop of the path walk                      -> ncclBroadcast
// librccl is opened at first use (dlopen), not linked: a process that never shards never loads it, and a process whose
// torch.distributed already loaded an RCCL keeps its own for torch.  RK_RCCL_LIB overrides the library name.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>          // types and prototypes only (decltype below); no symbol of it is referenced at link time
#else
// A ROCm install without the RCCL development headers still builds the library: the handful of declarations this file
// needs, as RCCL's public header states them (the library itself is looked up at run time either way).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclBroadcast(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
const char *ncclGetErrorString(ncclResult_t result);
}
#endif

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/rubiks_hip.h"
#include "rk_error.h"

using namespace rk;

namespace {

struct Rccl {
	void *lib = nullptr;
	decltype(&ncclGetUniqueId) get_unique_id = nullptr;
	decltype(&ncclCommInitRank) comm_init_rank = nullptr;
	decltype(&ncclCommDestroy) comm_destroy = nullptr;
	decltype(&ncclAllGather) all_gather = nullptr;
	decltype(&ncclBroadcast) broadcast = nullptr;
	decltype(&ncclSend) send = nullptr;
	decltype(&ncclRecv) recv = nullptr;
	decltype(&ncclGroupStart) group_start = nullptr;
	decltype(&ncclGroupEnd) group_end = nullptr;
	decltype(&ncclGetErrorString) error_string = nullptr;
	const char *why = nullptr;
} g;
std::once_flag g_once;

void load_rccl()
{
	const char *names[] = {std::getenv("RK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char *n : names) {
		if (n == nullptr || *n == 0) continue;
		g.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (g.lib != nullptr) break;
	}
	if (g.lib == nullptr) { g.why = "librccl.so.1 not found (set RK_RCCL_LIB)"; return; }
	#define RK_SYM(field, name) g.field = reinterpret_cast<decltype(g.field)>(dlsym(g.lib, name)); if (g.field == nullptr) { g.why = "missing RCCL symbol " name; return; }
	RK_SYM(get_unique_id, "ncclGetUniqueId") RK_SYM(comm_init_rank, "ncclCommInitRank") RK_SYM(comm_destroy, "ncclCommDestroy")
	RK_SYM(all_gather, "ncclAllGather") RK_SYM(broadcast, "ncclBroadcast") RK_SYM(send, "ncclSend") RK_SYM(recv, "ncclRecv")
	RK_SYM(group_start, "ncclGroupStart") RK_SYM(group_end, "ncclGroupEnd") RK_SYM(error_string, "ncclGetErrorString")
	#undef RK_SYM
}

int need_rccl()
{
	std::call_once(g_once, load_rccl);
	if (g.why != nullptr) return fail(RK_EHIP, "rk_comm: %s", g.why);
	return RK_OK;
}

#define RK_NCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return fail(RK_EHIP, "rk_comm: %s -> %s", #call, g.error_string(r_)); } while (0)

}  // namespace

struct rk_comm {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1;
};

extern "C" {

int rk_comm_unique_id(void *out_128_bytes)
{
	if (!out_128_bytes) return fail(RK_EINVAL, "rk_comm_unique_id: null pointer");
	if (int e = need_rccl()) return e;
	static_assert(sizeof(ncclUniqueId) == RK_COMM_ID_BYTES, "RCCL's unique id is 128 bytes");
	ncclUniqueId id;
	RK_NCCL(g.get_unique_id(&id));
	std::memcpy(out_128_bytes, &id, sizeof id);
	return RK_OK;
}

int rk_comm_create(rk_comm_t **out, const void *id_128_bytes, int rank, int world)
{
	if (!out || !id_128_bytes) return fail(RK_EINVAL, "rk_comm_create: null pointer");
	if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(RK_EINVAL, "rk_comm_create: rank %d / world %d out of range", rank, world);
	if (int e = need_rccl()) return e;
	int dev = -1;
	if (hipGetDevice(&dev) != hipSuccess) return fail(RK_EHIP, "rk_comm_create: no HIP device selected (rk_init first)");
	ncclUniqueId id;
	std::memcpy(&id, id_128_bytes, sizeof id);
	rk_comm *c = new rk_comm();
	c->rank = rank; c->world = world;
	ncclResult_t r = g.comm_init_rank(&c->comm, world, id, rank);          // collective: every rank calls it with the same id
	if (r != ncclSuccess) { delete c; return fail(RK_EHIP, "rk_comm_create: ncclCommInitRank -> %s", g.error_string(r)); }
	*out = c;
	return RK_OK;
}

int rk_comm_destroy(rk_comm_t *c)
{
	if (!c) return RK_OK;
	if (c->comm != nullptr && g.comm_destroy != nullptr) (void)g.comm_destroy(c->comm);
	delete c;
	return RK_OK;
}

int rk_comm_rank(const rk_comm_t *c) { return c ? c->rank : -1; }
int rk_comm_world(const rk_comm_t *c) { return c ? c->world : 0; }

int rk_comm_all_gather(rk_comm_t *c, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream)
{
	if (!c || !d_send || !d_recv) return fail(RK_EINVAL, "rk_comm_all_gather: null argument");
	RK_NCCL(g.all_gather(d_send, d_recv, bytes_per_rank, ncclChar, c->comm, (hipStream_t)stream));
	return RK_OK;
}

int rk_comm_all_to_all(rk_comm_t *c, const void *d_send, void *d_recv, size_t bytes_per_peer, void *stream)
{
	if (!c || !d_send || !d_recv) return fail(RK_EINVAL, "rk_comm_all_to_all: null argument");
	if (d_send == d_recv) return fail(RK_EINVAL, "rk_comm_all_to_all: send and receive buffers must differ");
	const char *s = static_cast<const char *>(d_send);
	char *r = static_cast<char *>(d_recv);
	// Between group_start and group_end nothing may return: RCCL's group depth is thread-local, and a raised depth makes every
	// later collective of this thread wait for a group_end that never comes.  The first error is remembered, the group is
	// always closed, then the error is reported.
	RK_NCCL(g.group_start());
	ncclResult_t first = ncclSuccess;
	const char *what = "";
	for (int p = 0; p < c->world && first == ncclSuccess; p++) {
		first = g.send(s + (size_t)p * bytes_per_peer, bytes_per_peer, ncclChar, p, c->comm, (hipStream_t)stream);
		what = "ncclSend";
		if (first != ncclSuccess) break;
		first = g.recv(r + (size_t)p * bytes_per_peer, bytes_per_peer, ncclChar, p, c->comm, (hipStream_t)stream);
		what = "ncclRecv";
	}
	const ncclResult_t end = g.group_end();
	if (first != ncclSuccess) return fail(RK_EHIP, "rk_comm_all_to_all: %s -> %s", what, g.error_string(first));
	if (end != ncclSuccess) return fail(RK_EHIP, "rk_comm_all_to_all: ncclGroupEnd -> %s", g.error_string(end));
	return RK_OK;
}

int rk_comm_broadcast(rk_comm_t *c, void *d_buf, size_t bytes, int root, void *stream)
{
	if (!c || !d_buf) return fail(RK_EINVAL, "rk_comm_broadcast: null argument");
	if (root < 0 || root >= c->world) return fail(RK_EINVAL, "rk_comm_broadcast: root %d outside the communicator", root);
	RK_NCCL(g.broadcast(d_buf, d_buf, bytes, ncclChar, root, c->comm, (hipStream_t)stream));
	return RK_OK;
}

}  // extern "C"
