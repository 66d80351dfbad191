// rccl_shim.cpp — TEST HARNESS: the eight RCCL entry points sharded.cpp binds (dlopen via JVECTOR_HIP_RCCL_PATH),
// implemented over POSIX shared memory for `world` PROCESSES on one host whose "device memory" is host memory (the mock
// HIP runtime of tests/mock).  It lets the world_size-2 CPU test drive the C ABI's multi-rank path — rendezvous by unique id,
// grouped all-gathers, the gathered layout — without a GPU.  Never loaded by the product.
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace {
constexpr size_t kSlot = 8u << 20;  // bytes per rank per all-gather
struct Shared {
    std::atomic<int> ready;
    pthread_barrier_t barrier;
    char data[1];
};
struct Comm {
    Shared *sh;
    size_t bytes;
    int rank, world;
    char name[64];
};
size_t dt_size(int dt) { return dt == 0 || dt == 1 ? 1 : (dt == 4 || dt == 5 || dt == 8) ? 8 : (dt == 6 || dt == 9) ? 2 : 4; }
}  // namespace

extern "C" {

int ncclGetUniqueId(char *id)
{
    static std::atomic<int> seq{0};
    memset(id, 0, 128);
    snprintf(id, 128, "/jv_rccl_shim_%d_%d", (int)getpid(), seq.fetch_add(1));
    return 0;
}

struct IdByValue {
    char internal[128];
};

int ncclCommInitRank(void **out, int world, IdByValue id, int rank)
{
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->bytes = sizeof(Shared) + kSlot * (size_t)world;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
    } else {
        for (int i = 0; i < 20000 && fd < 0; ++i) {
            fd = shm_open(c->name, O_RDWR, 0600);
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) return 2;
        for (int i = 0; i < 20000; ++i) {  // wait for rank 0's ftruncate
            off_t sz = lseek(fd, 0, SEEK_END);
            if ((size_t)sz >= c->bytes) break;
            usleep(1000);
        }
    }
    c->sh = (Shared *)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (c->sh == MAP_FAILED) return 2;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->sh->barrier, &a, (unsigned)world);
        c->sh->ready.store(1);
    } else {
        while (c->sh->ready.load() != 1) usleep(200);
    }
    pthread_barrier_wait(&c->sh->barrier);
    *out = c;
    return 0;
}

int ncclCommDestroy(void *comm)
{
    Comm *c = (Comm *)comm;
    pthread_barrier_wait(&c->sh->barrier);
    munmap(c->sh, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

int ncclAllGather(const void *send, void *recv, size_t count, int dt, void *comm, void * /*stream*/)
{
    Comm *c = (Comm *)comm;
    const size_t bytes = count * dt_size(dt);
    if (bytes > kSlot) return 5;
    memcpy(c->sh->data + kSlot * (size_t)c->rank, send, bytes);
    pthread_barrier_wait(&c->sh->barrier);
    for (int r = 0; r < c->world; ++r) memcpy((char *)recv + bytes * (size_t)r, c->sh->data + kSlot * (size_t)r, bytes);
    pthread_barrier_wait(&c->sh->barrier);
    return 0;
}

int ncclCommCount(void *comm, int *count)
{
    *count = ((Comm *)comm)->world;
    return 0;
}

int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 5 ? "invalid argument (message larger than the shim's slot)" : "shim system error"; }

}  // extern "C"
