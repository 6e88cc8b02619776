/* TEST FIXTURE, not a decoder and not part of the product: a backend in plain C
 * behind the library's plug-in boundary (hwang_backend_vtable, include/hwang_amd.h).
 * It emits one "frame" per fed sample whose first four bytes are the sample's
 * first four payload bytes (rest zero), which lets the DecoderAutomata /
 * ParallelRetriever protocol be pinned on machines without a GPU - the way
 * SURVEY.md Appendix B-3 drove the reference automata with a counting fake
 * backend.  tests/conftest.py builds it into tests/_build/libstamp_backend.so
 * and registers it as decoder type 100.  Call counters are process-wide. */
#include "../../include/hwang_amd.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  pthread_mutex_t m;
  uint32_t width, height;
  uint32_t* q; /* ring of stamps */
  size_t cap, head, count;
  char error[96];
} Stamp;

/* configures, feeds, keyframe_feeds, empty_feeds, flushes, gets, discards, hints */
static int32_t g_trace[8];
static pthread_mutex_t g_trace_m = PTHREAD_MUTEX_INITIALIZER;
static void bump(int k) {
  pthread_mutex_lock(&g_trace_m);
  ++g_trace[k];
  pthread_mutex_unlock(&g_trace_m);
}
void hwang_stamp_trace(int32_t out[8]) {
  pthread_mutex_lock(&g_trace_m);
  memcpy(out, g_trace, sizeof(g_trace));
  pthread_mutex_unlock(&g_trace_m);
}
void hwang_stamp_reset(void) {
  pthread_mutex_lock(&g_trace_m);
  memset(g_trace, 0, sizeof(g_trace));
  pthread_mutex_unlock(&g_trace_m);
}

static void* st_create(int device_type, int32_t device_id, uint32_t num_devices) {
  (void)device_type; (void)device_id; (void)num_devices;
  Stamp* s = (Stamp*)calloc(1, sizeof(Stamp));
  if (!s) return NULL;
  pthread_mutex_init(&s->m, NULL);
  s->cap = 1024;
  s->q = (uint32_t*)malloc(s->cap * sizeof(uint32_t));
  return s;
}
static void st_destroy(void* p) {
  Stamp* s = (Stamp*)p;
  pthread_mutex_destroy(&s->m);
  free(s->q);
  free(s);
}
static int st_configure(void* p, uint32_t w, uint32_t h, const char* format, const uint8_t* extra, size_t n) {
  Stamp* s = (Stamp*)p;
  (void)format; (void)extra; (void)n;
  pthread_mutex_lock(&s->m);
  s->width = w;
  s->height = h;
  pthread_mutex_unlock(&s->m);
  bump(0);
  return 0;
}
static int st_feed(void* p, const uint8_t* data, size_t size, int keyframe) {
  Stamp* s = (Stamp*)p;
  if (size == 0) {
    bump(3);
    return 0;
  }
  bump(1);
  if (keyframe) bump(2);
  uint32_t stamp = 0;
  memcpy(&stamp, data, size < 4 ? size : 4);
  pthread_mutex_lock(&s->m);
  if (s->count == s->cap) { /* grow, unrolling the ring */
    uint32_t* nq = (uint32_t*)malloc(2 * s->cap * sizeof(uint32_t));
    for (size_t i = 0; i < s->count; ++i) nq[i] = s->q[(s->head + i) % s->cap];
    free(s->q);
    s->q = nq;
    s->head = 0;
    s->cap *= 2;
  }
  s->q[(s->head + s->count++) % s->cap] = stamp;
  pthread_mutex_unlock(&s->m);
  return 0;
}
static int st_flush(void* p) {
  (void)p;
  bump(4);
  return 0;
}
static int st_discard(void* p) {
  Stamp* s = (Stamp*)p;
  int popped = 0;
  pthread_mutex_lock(&s->m);
  if (s->count) {
    s->head = (s->head + 1) % s->cap;
    --s->count;
    popped = 1;
  }
  pthread_mutex_unlock(&s->m);
  if (popped) bump(6);
  return 0;
}
static int st_get(void* p, uint8_t* dst, size_t dst_size) {
  Stamp* s = (Stamp*)p;
  pthread_mutex_lock(&s->m);
  if (!s->count) {
    pthread_mutex_unlock(&s->m);
    return 0;
  }
  const size_t need = (size_t)s->width * s->height * 3;
  if (dst_size < need || need < 4) {
    strcpy(s->error, "StampVideoDecoder::get_frame: buffer too small");
    pthread_mutex_unlock(&s->m);
    return 1;
  }
  memset(dst, 0, need);
  memcpy(dst, &s->q[s->head], 4);
  s->head = (s->head + 1) % s->cap;
  --s->count;
  pthread_mutex_unlock(&s->m);
  bump(5);
  return 0;
}
static int st_buffered(void* p) {
  Stamp* s = (Stamp*)p;
  pthread_mutex_lock(&s->m);
  const int n = (int)s->count;
  pthread_mutex_unlock(&s->m);
  return n;
}
static int st_wait(void* p) {
  (void)p;
  return 0;
}
static const char* st_error(void* p) { return ((Stamp*)p)->error; }
static void st_wanted(void* p, const uint64_t* ordinals, size_t n) {
  (void)p; (void)ordinals; (void)n;
  bump(7);
}

const hwang_backend_vtable hwang_stamp_vtable = {st_create, st_destroy, st_configure, st_feed,  st_flush,  st_discard,
                                                 st_get,    st_buffered, st_wait,     st_error, st_wanted};
