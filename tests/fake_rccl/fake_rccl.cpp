// fake_rccl.cpp -- TEST DOUBLE of the four RCCL entry points libvinsat_ba.so resolves at run time (vba_sh_comm_init names the
// library by path): ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy (+ ncclGetErrorString), carried by a
// POSIX shared-memory segment on the host.  RCCL wants one device per rank; a one-GPU box can therefore never run the
// library-issued exchanges of the observation-sharded mode with more than one rank -- with this stand-in two rank PROCESSES on
// the one GPU execute vba_sh_call with world = 2: slot order, +inf padding of unequal shards, rank-ordered reduce, id hand-over.
// ncclAllGather here = wait for the stream, copy the send buffer to this rank's slot, barrier, copy every slot back, barrier.
// Test infrastructure only (tests/test_sharded_fake_rccl_gpu.py builds it with g++); never shipped, never used by the product.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr size_t kSlotBytes = 8u << 20;         // per rank and exchange: far above anything the tests gather
constexpr int kMaxRanks = 8;

struct Shared {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    int nranks;
    alignas(64) unsigned char slots[1];         // [nranks][kSlotBytes]
};

struct FakeComm {
    Shared* sh = nullptr;
    size_t bytes = 0;
    int nranks = 0, rank = 0;
    char name[64] = {};
};

bool barrier(Shared* sh, int nranks) {          // sense-reversing, bounded: a rank that never arrives fails the test instead of hanging it
    const int gen = sh->generation.load(std::memory_order_acquire);
    if (sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
        sh->arrived.store(0, std::memory_order_relaxed);
        sh->generation.store(gen + 1, std::memory_order_release);
        return true;
    }
    const time_t t0 = time(nullptr);
    while (sh->generation.load(std::memory_order_acquire) == gen) {
        usleep(20);
        if (time(nullptr) - t0 > 60) return false;
    }
    return true;
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%ld", (int)getpid(), (long)time(nullptr) ^ (long)clock());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    FakeComm* c = new FakeComm();
    c->nranks = nranks;
    c->rank = rank;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->bytes = sizeof(Shared) + (size_t)nranks * kSlotBytes;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
    } else {
        const time_t t0 = time(nullptr);
        struct stat st;
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < c->bytes) {
            if (fd >= 0) { close(fd); fd = -1; }
            usleep(200);
            if (time(nullptr) - t0 > 60) { delete c; return ncclSystemError; }
        }
    }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared*>(p);
    if (rank == 0) c->sh->nranks = nranks;      // (a fresh segment is zero filled: counters start at 0)
    c->sh->attached.fetch_add(1);
    const time_t t0 = time(nullptr);
    while (c->sh->attached.load() < nranks) {   // collective: returns when every rank has joined
        usleep(200);
        if (time(nullptr) - t0 > 60) return ncclSystemError;
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
    const size_t bytes = sendcount * type_size(datatype);
    if (bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->slots + (size_t)c->rank * kSlotBytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c->sh, c->nranks)) return ncclSystemError;
    for (int r = 0; r < c->nranks; ++r)
        if (hipMemcpy(static_cast<char*>(recvbuff) + (size_t)r * bytes, c->sh->slots + (size_t)r * kSlotBytes, bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    if (!barrier(c->sh, c->nranks)) return ncclSystemError;     // the slots may be overwritten again
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c) return ncclSuccess;
    if (c->sh) munmap(c->sh, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
        case ncclSystemError: return "fake_rccl: shared memory / barrier failed (a rank did not arrive within 60 s)";
        case ncclInvalidArgument: return "fake_rccl: invalid argument";
        default: return "fake_rccl: error";
    }
}

}  // extern "C"
