/* mercury_shm.h — the transport on the OUTPUT side of the RX path.
 *
 * Mercury hands decoded payloads to client programs through a blocking byte ring buffer in POSIX shared
 * memory (RX_SHM mode, source/physical_layer/telecom_system.cc:2326-2333; client: examples/receiver.c).
 * This is a from-scratch implementation of that transport with the SAME shared-memory layout and
 * protocol, so an unmodified reference client (or a reference writer) interoperates with it:
 *
 *   object "<base_name>-1"   the data bytes, `size` of them                      (ring_buffer_posix.cc:138-147)
 *   object "<base_name>-2"   struct { size_t head, tail, max; bool full;          (include/common/ring_buffer_posix.h:37-50)
 *                                     pthread_mutex_t mutex; pthread_cond_t cond; }  both PTHREAD_PROCESS_SHARED
 *   head = next write index, tail = next read index, full disambiguates head == tail; a writer blocks on
 *   `cond` until `len` bytes are free, a reader until data is present, each signals `cond` after moving
 *   its index (write_buffer / read_buffer / read_buffer_all, ring_buffer_posix.cc:462-631).
 *
 * Reference defaults: base_name "/mercury-comm", size 131072 (include/common/common_defines.h:208-209).
 * Every function returns MGPU_OK (0) or an MGPU_ERR_* code from mercury_gpu.h unless stated otherwise.
 */
#ifndef MERCURY_SHM_H
#define MERCURY_SHM_H

#include <stddef.h>
#include <stdint.h>

#include "mercury_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_SHM_PAYLOAD_NAME "/mercury-comm"     /* SHM_PAYLOAD_NAME */
#define MGPU_SHM_PAYLOAD_BUFFER_SIZE 131072       /* SHM_PAYLOAD_BUFFER_SIZE */

typedef struct mgpu_shm mgpu_shm;

/* circular_buf_init_shm (ring_buffer_posix.cc:126-216): (re)create both objects, initialise the lock, empty ring */
int mgpu_shm_create(const char* base_name, size_t size, mgpu_shm** out);
/* circular_buf_connect_shm (:218-257): attach to a ring somebody created; MGPU_ERR_ARG if absent or of another size */
int mgpu_shm_connect(const char* base_name, size_t size, mgpu_shm** out);
/* circular_buf_free_shm: drop this process's handle (the objects stay) */
void mgpu_shm_close(mgpu_shm* ring);
/* circular_buf_destroy_shm (:264-282): unmap and unlink both objects, then drop the handle */
void mgpu_shm_destroy(mgpu_shm* ring);

size_t mgpu_shm_used(mgpu_shm* ring);             /* size_buffer */
size_t mgpu_shm_free(mgpu_shm* ring);             /* circular_buf_free_size */
size_t mgpu_shm_capacity(mgpu_shm* ring);         /* circular_buf_capacity */
void mgpu_shm_clear(mgpu_shm* ring);              /* clear_buffer / circular_buf_reset */

/* write_buffer: blocks until len bytes are free. len must be <= capacity. */
int mgpu_shm_write(mgpu_shm* ring, const uint8_t* data, size_t len);
/* read_buffer: blocks until len bytes are present */
int mgpu_shm_read(mgpu_shm* ring, uint8_t* data, size_t len);
/* read_buffer_all: blocks until anything is present, takes everything; returns the byte count (>0) or -1 */
long mgpu_shm_read_all(mgpu_shm* ring, uint8_t* data);

/* The publishing step of RX_SHM_process_main for a batch (telecom_system.cc:2326-2333): for every frame with
 * stats[f].message_decoded, write its first payload_bytes bytes if that much is free, otherwise count it as lost
 * ("Decoded frame lost because of full buffer!"). Never blocks. published / lost may be NULL. */
int mgpu_shm_publish_decoded(mgpu_shm* ring, const uint8_t* payload, const mgpu_frame_stats* stats, int F,
                             int payload_stride, int payload_bytes, int* published, int* lost);

#ifdef __cplusplus
}
#endif
#endif /* MERCURY_SHM_H */
