// POSIX shared-memory byte ring on the output side of the RX path — see include/mercury_shm.h for the layout
// and the reference code it interoperates with (source/common/ring_buffer_posix.cc, shm_posix.cc).
#include "../../include/mercury_shm.h"

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <new>
#include <string>

namespace {

// must stay layout-compatible with struct circular_buf_t_aux (include/common/ring_buffer_posix.h:37-50)
struct RingState {
    size_t head, tail, max;
    bool full;
    pthread_mutex_t mutex;
    pthread_cond_t cond;
};

constexpr size_t kMaxName = 255;   // MAX_POSIX_SHM_NAME, include/common/shm_posix.h:27

struct Lock {
    pthread_mutex_t* m;
    explicit Lock(pthread_mutex_t* mm) : m(mm) { pthread_mutex_lock(m); }
    ~Lock() { pthread_mutex_unlock(m); }
};

size_t used_locked(const RingState* s) {
    if (s->full) return s->max;
    return s->head >= s->tail ? s->head - s->tail : s->max + s->head - s->tail;
}
size_t free_locked(const RingState* s) { return s->max - used_locked(s); }

void* map_object(const std::string& name, size_t size, bool create) {
    int fd;
    if (create) {
        // shm_create_and_get_fd (shm_posix.cc:66-140): an existing object is replaced
        shm_unlink(name.c_str());
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0644);
        if (fd < 0) return nullptr;
        if (ftruncate(fd, off_t(size)) != 0) { close(fd); shm_unlink(name.c_str()); return nullptr; }
    } else {
        fd = shm_open(name.c_str(), O_RDWR, 0644);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || size_t(st.st_size) < size) { close(fd); return nullptr; }
    }
    void* p = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}

}  // namespace

struct mgpu_shm {
    RingState* state = nullptr;
    uint8_t* bytes = nullptr;
    size_t size = 0;
    std::string base;
};

namespace {

int open_ring(const char* base_name, size_t size, bool create, mgpu_shm** out) {
    if (!base_name || !out || size == 0 || std::strlen(base_name) + 2 >= kMaxName) return MGPU_ERR_ARG;
    *out = nullptr;
    mgpu_shm* r = new (std::nothrow) mgpu_shm();
    if (!r) return MGPU_ERR_DEVICE;
    r->base = base_name;
    r->size = size;
    r->bytes = static_cast<uint8_t*>(map_object(r->base + "-1", size, create));
    r->state = static_cast<RingState*>(map_object(r->base + "-2", sizeof(RingState), create));
    if (!r->bytes || !r->state) {
        if (r->bytes) munmap(r->bytes, size);
        if (r->state) munmap(r->state, sizeof(RingState));
        if (create) { shm_unlink((r->base + "-1").c_str()); shm_unlink((r->base + "-2").c_str()); }
        delete r;
        return MGPU_ERR_ARG;
    }
    if (create) {
        r->state->max = size;
        pthread_mutexattr_t ma;
        pthread_mutexattr_init(&ma);
        pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
        pthread_condattr_t ca;
        pthread_condattr_init(&ca);
        pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
        pthread_mutex_init(&r->state->mutex, &ma);
        pthread_cond_init(&r->state->cond, &ca);
        pthread_mutexattr_destroy(&ma);
        pthread_condattr_destroy(&ca);
        Lock l(&r->state->mutex);
        r->state->head = r->state->tail = 0;
        r->state->full = false;
    } else if (r->state->max != size) {          // the reference asserts this (ring_buffer_posix.cc:254)
        munmap(r->bytes, size);
        munmap(r->state, sizeof(RingState));
        delete r;
        return MGPU_ERR_ARG;
    }
    *out = r;
    return MGPU_OK;
}

// copy in at head / out at tail with wrap-around; caller holds the lock and has checked the space
void put_locked(mgpu_shm* r, const uint8_t* data, size_t len) {
    RingState* s = r->state;
    const size_t first = len < s->max - s->head ? len : s->max - s->head;
    std::memcpy(r->bytes + s->head, data, first);
    std::memcpy(r->bytes, data + first, len - first);
    if (s->full) s->tail = (s->tail + len) % s->max;            // advance_pointer_n, ring_buffer_posix.cc:29-44
    s->head = (s->head + len) % s->max;
    s->full = len > 0 && s->head == s->tail;
}
void get_locked(mgpu_shm* r, uint8_t* data, size_t len) {
    RingState* s = r->state;
    const size_t first = len < s->max - s->tail ? len : s->max - s->tail;
    std::memcpy(data, r->bytes + s->tail, first);
    std::memcpy(data + first, r->bytes, len - first);
    if (len > 0) s->full = false;                                // retreat_pointer_n, :61-67
    s->tail = (s->tail + len) % s->max;
}

}  // namespace

extern "C" {

int mgpu_shm_create(const char* base_name, size_t size, mgpu_shm** out) { return open_ring(base_name, size, true, out); }
int mgpu_shm_connect(const char* base_name, size_t size, mgpu_shm** out) { return open_ring(base_name, size, false, out); }

void mgpu_shm_close(mgpu_shm* r) {
    if (!r) return;
    munmap(r->bytes, r->size);
    munmap(r->state, sizeof(RingState));
    delete r;
}

void mgpu_shm_destroy(mgpu_shm* r) {
    if (!r) return;
    const std::string base = r->base;
    mgpu_shm_close(r);
    shm_unlink((base + "-1").c_str());
    shm_unlink((base + "-2").c_str());
}

size_t mgpu_shm_used(mgpu_shm* r) { if (!r) return 0; Lock l(&r->state->mutex); return used_locked(r->state); }
size_t mgpu_shm_free(mgpu_shm* r) { if (!r) return 0; Lock l(&r->state->mutex); return free_locked(r->state); }
size_t mgpu_shm_capacity(mgpu_shm* r) { if (!r) return 0; Lock l(&r->state->mutex); return r->state->max; }
void mgpu_shm_clear(mgpu_shm* r) {
    if (!r) return;
    Lock l(&r->state->mutex);
    r->state->head = r->state->tail = 0;
    r->state->full = false;
}

int mgpu_shm_write(mgpu_shm* r, const uint8_t* data, size_t len) {
    if (!r || (!data && len) || len > r->size) return MGPU_ERR_ARG;
    Lock l(&r->state->mutex);
    while (free_locked(r->state) < len) pthread_cond_wait(&r->state->cond, &r->state->mutex);
    put_locked(r, data, len);
    pthread_cond_signal(&r->state->cond);
    return MGPU_OK;
}

int mgpu_shm_read(mgpu_shm* r, uint8_t* data, size_t len) {
    if (!r || !data || len > r->size) return MGPU_ERR_ARG;
    Lock l(&r->state->mutex);
    while (used_locked(r->state) < len) pthread_cond_wait(&r->state->cond, &r->state->mutex);
    get_locked(r, data, len);
    pthread_cond_signal(&r->state->cond);
    return MGPU_OK;
}

long mgpu_shm_read_all(mgpu_shm* r, uint8_t* data) {
    if (!r || !data) return -1;
    Lock l(&r->state->mutex);
    size_t len;
    while ((len = used_locked(r->state)) == 0) pthread_cond_wait(&r->state->cond, &r->state->mutex);
    get_locked(r, data, len);
    pthread_cond_signal(&r->state->cond);
    return long(len);
}

int mgpu_shm_publish_decoded(mgpu_shm* r, const uint8_t* payload, const mgpu_frame_stats* stats, int F, int payload_stride,
                             int payload_bytes, int* published, int* lost) {
    if (!r || !payload || !stats || F < 0 || payload_bytes < 0 || payload_stride < payload_bytes || size_t(payload_bytes) > r->size)
        return MGPU_ERR_ARG;
    int np = 0, nl = 0;
    for (int f = 0; f < F; ++f) {
        if (!stats[f].message_decoded) continue;                 // telecom_system.cc:2317
        Lock l(&r->state->mutex);
        if (size_t(payload_bytes) <= free_locked(r->state)) {    // :2326-2329
            put_locked(r, payload + size_t(f) * payload_stride, size_t(payload_bytes));
            pthread_cond_signal(&r->state->cond);
            ++np;
        } else {
            ++nl;
        }
    }
    if (published) *published = np;
    if (lost) *lost = nl;
    return MGPU_OK;
}

}  // extern "C"
