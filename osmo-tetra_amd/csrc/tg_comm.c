/*
 * tg_comm.c -- the one exchange of the multi-GPU path in C: the gather of decoded blocks (wire records) to the
 * collecting rank, over RCCL (BASELINE.json north star: "RCCL only for the final decoded-block gather over xGMI";
 * SURVEY.md 8(e)).  One process per GPU; channels are sharded, decoding needs no collective.
 *
 * RCCL is loaded on first use (dlopen), so libtetra_gpu.so itself does not depend on it: a process that already
 * holds an RCCL (a PyTorch process does) gets that one, otherwise ROCm's librccl.so.1.  The gather is a grouped
 * send / receive: every rank sends its block to the root, the root posts one receive per rank -- each peer -> root
 * transfer rides its own xGMI link, nothing is relayed.
 *
 * The reference has no counterpart (it runs one process per channel and prints).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

/* The handful of NCCL / RCCL declarations this file needs (the published, ABI-stable NCCL 2 interface; rccl.h itself
 * pulls in the C++ HIP runtime header and cannot be included from C) */
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
#define ncclSuccess 0
typedef int ncclDataType_t;
#define ncclUint8 1
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);

#include "tetra_gpu.h"
#include "tg_internal.h"

struct tgpu_comm {
	struct tgpu_engine *eng;
	ncclComm_t comm;
	int rank, world;
};

static struct {
	void *lib;
	__typeof__(&ncclGetUniqueId) get_unique_id;
	__typeof__(&ncclCommInitRank) comm_init_rank;
	__typeof__(&ncclCommDestroy) comm_destroy;
	__typeof__(&ncclGroupStart) group_start;
	__typeof__(&ncclGroupEnd) group_end;
	__typeof__(&ncclSend) send;
	__typeof__(&ncclRecv) recv;
} rc;

static pthread_mutex_t rc_lock = PTHREAD_MUTEX_INITIALIZER;

static int rccl_load_locked(void);

static int rccl_load(void)
{
	pthread_mutex_lock(&rc_lock);
	const int r = rccl_load_locked();
	pthread_mutex_unlock(&rc_lock);
	return r;
}

static int rccl_load_locked(void)
{
	if (rc.lib)
		return TGPU_OK;
	static const char *const names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
	void *h = NULL;
	for (unsigned i = 0; i < 2 && !h; i++)		/* an RCCL this process already holds */
		h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
	for (unsigned i = 0; i < sizeof(names) / sizeof(names[0]) && !h; i++)
		h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
	if (!h)
		return TGPU_ENOSYS;
#define SYM(field, name) do { *(void **)&rc.field = dlsym(h, name); if (!rc.field) { dlclose(h); return TGPU_ENOSYS; } } while (0)
	SYM(get_unique_id, "ncclGetUniqueId");
	SYM(comm_init_rank, "ncclCommInitRank");
	SYM(comm_destroy, "ncclCommDestroy");
	SYM(group_start, "ncclGroupStart");
	SYM(group_end, "ncclGroupEnd");
	SYM(send, "ncclSend");
	SYM(recv, "ncclRecv");
#undef SYM
	rc.lib = h;
	return TGPU_OK;
}

int tgpu_comm_unique_id(uint8_t id[TGPU_COMM_ID_BYTES])
{
	_Static_assert(TGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
	if (!id)
		return TGPU_EINVAL;
	int r = rccl_load();
	if (r)
		return r;
	ncclUniqueId u;
	if (rc.get_unique_id(&u) != ncclSuccess)
		return TGPU_ECOMM;
	memcpy(id, u.internal, TGPU_COMM_ID_BYTES);
	return TGPU_OK;
}

int tgpu_comm_create(struct tgpu_engine *eng, const uint8_t id[TGPU_COMM_ID_BYTES], int rank, int world, struct tgpu_comm **out)
{
	if (!eng || !id || !out || world < 1 || rank < 0 || rank >= world)
		return TGPU_EINVAL;
	int r = rccl_load();
	if (r)
		return r;
	if ((r = tgpi_engine_bind(eng)))
		return r;
	struct tgpu_comm *c = calloc(1, sizeof(*c));
	if (!c)
		return TGPU_ENOMEM;
	ncclUniqueId u;
	memcpy(u.internal, id, TGPU_COMM_ID_BYTES);
	if (rc.comm_init_rank(&c->comm, world, u, rank) != ncclSuccess) {
		free(c);
		return TGPU_ECOMM;
	}
	c->eng = eng;
	c->rank = rank;
	c->world = world;
	*out = c;
	return TGPU_OK;
}

int tgpu_comm_gather(struct tgpu_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root, void *hip_stream)
{
	if (!c || !d_send || root < 0 || root >= c->world || (c->rank == root && !d_recv))
		return TGPU_EINVAL;
	if (!nbytes)
		return TGPU_OK;
	int r = tgpi_engine_bind(c->eng);
	if (r)
		return r;
	hipStream_t s = (hipStream_t)hip_stream;
	ncclResult_t e = rc.group_start();
	if (e == ncclSuccess && c->rank == root)
		for (int p = 0; p < c->world && e == ncclSuccess; p++)
			e = rc.recv((uint8_t *)d_recv + (size_t)p * nbytes, nbytes, ncclUint8, p, c->comm, s);
	if (e == ncclSuccess)
		e = rc.send(d_send, nbytes, ncclUint8, root, c->comm, s);
	ncclResult_t e2 = rc.group_end();
	return (e == ncclSuccess && e2 == ncclSuccess) ? TGPU_OK : TGPU_ECOMM;
}

int tgpu_comm_gatherv(struct tgpu_comm *c, const void *d_send, const size_t *nbytes, void *d_recv, const size_t *offs, int root,
		      void *hip_stream)
{
	if (!c || !nbytes || root < 0 || root >= c->world || (c->rank == root && (!d_recv || !offs)))
		return TGPU_EINVAL;
	if (nbytes[c->rank] && !d_send)
		return TGPU_EINVAL;
	int r = tgpi_engine_bind(c->eng);
	if (r)
		return r;
	hipStream_t s = (hipStream_t)hip_stream;
	/* a rank with nothing to send takes no part in the group (its zero is known to the root as well) */
	ncclResult_t e = rc.group_start();
	if (e == ncclSuccess && c->rank == root)
		for (int p = 0; p < c->world && e == ncclSuccess; p++)
			if (nbytes[p])
				e = rc.recv((uint8_t *)d_recv + offs[p], nbytes[p], ncclUint8, p, c->comm, s);
	if (e == ncclSuccess && nbytes[c->rank])
		e = rc.send(d_send, nbytes[c->rank], ncclUint8, root, c->comm, s);
	ncclResult_t e2 = rc.group_end();
	return (e == ncclSuccess && e2 == ncclSuccess) ? TGPU_OK : TGPU_ECOMM;
}

int tgpu_comm_gatherv_batch(struct tgpu_comm *c, int nmsg, const void *const *d_send, const size_t *nbytes, void *d_recv,
			    const size_t *offs, int root, void *hip_stream)
{
	if (!c || nmsg < 0 || (nmsg && (!d_send || !nbytes)) || root < 0 || root >= c->world || (nmsg && c->rank == root && (!d_recv || !offs)))
		return TGPU_EINVAL;
	if (!nmsg)
		return TGPU_OK;
	for (int m = 0; m < nmsg; m++)
		if (nbytes[(size_t)m * c->world + c->rank] && !d_send[m])
			return TGPU_EINVAL;
	int r = tgpi_engine_bind(c->eng);
	if (r)
		return r;
	hipStream_t s = (hipStream_t)hip_stream;
	/* ONE group for all the messages: the sends and receives between a pair of ranks are matched in the order they are
	 * issued (message by message on both sides), and the group is one launch on every rank */
	ncclResult_t e = rc.group_start();
	for (int m = 0; m < nmsg && e == ncclSuccess; m++) {
		const size_t *nb = nbytes + (size_t)m * c->world;
		if (c->rank == root)
			for (int p = 0; p < c->world && e == ncclSuccess; p++)
				if (nb[p])
					e = rc.recv((uint8_t *)d_recv + offs[(size_t)m * c->world + p], nb[p], ncclUint8, p, c->comm, s);
		if (e == ncclSuccess && nb[c->rank])
			e = rc.send(d_send[m], nb[c->rank], ncclUint8, root, c->comm, s);
	}
	ncclResult_t e2 = rc.group_end();
	return (e == ncclSuccess && e2 == ncclSuccess) ? TGPU_OK : TGPU_ECOMM;
}

void tgpu_comm_destroy(struct tgpu_comm *c)
{
	if (!c)
		return;
	if (rc.lib && c->comm)
		rc.comm_destroy(c->comm);
	free(c);
}
