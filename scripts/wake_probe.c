// wake_probe.c - how often does a thread that sleeps and wakes like a host rANS worker (wake, code a table piece for ~0.3 ms,
// sleep until the next piece lands) lose milliseconds to OTHER tenants' runnable threads on a shared host - and does a short
// scheduler slice (sched_setattr sched_runtime, EEVDF custom slices + PREEMPT_SHORT, Linux >= 6.12) help it?   (no GPU involved)
//
//   wake_probe <threads> <seconds> <slice_us: 0 = the default>
//
// Every thread: busy `work` us, then clock_nanosleep `nap` us; logged: wake-ups > 0.5 ms late, and gaps > 0.5 ms inside the busy
// phase (the thread was preempted).  Output: totals per run; the worst events.
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

struct sched_attr_ {
  uint32_t size, sched_policy;
  uint64_t sched_flags;
  int32_t sched_nice;
  uint32_t sched_priority;
  uint64_t sched_runtime, sched_deadline, sched_period;
  uint32_t sched_util_min, sched_util_max;
};

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

typedef struct {
  double t_end, t0;
  int slice_us, set_rc;
  int late, cut, wakes;
  double late_ms, cut_ms, worst;
} W;

static void *worker(void *arg) {
  W *w = (W *)arg;
  if (w->slice_us > 0) {
    struct sched_attr_ a;
    memset(&a, 0, sizeof a);
    a.size = sizeof a;
    a.sched_policy = 0; // SCHED_OTHER
    a.sched_runtime = (uint64_t)w->slice_us * 1000;
    w->set_rc = syscall(SYS_sched_setattr, 0, &a, 0) ? errno : 0;
  }
  while (now_ms() < w->t_end) {
    // busy 300 us
    double last = now_ms();
    const double until = last + 0.3;
    while (last < until) {
      const double t = now_ms();
      if (t - last > 0.5) ++w->cut, w->cut_ms += t - last, w->worst = t - last > w->worst ? t - last : w->worst;
      last = t;
    }
    // nap 200 us
    struct timespec nap = {0, 200000};
    const double t0 = now_ms();
    clock_nanosleep(CLOCK_MONOTONIC, 0, &nap, NULL);
    const double late = now_ms() - t0 - 0.2;
    ++w->wakes;
    if (late > 0.5) ++w->late, w->late_ms += late, w->worst = late > w->worst ? late : w->worst;
  }
  return NULL;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16;
  const double secs = argc > 2 ? atof(argv[2]) : 2.0;
  const int slice = argc > 3 ? atoi(argv[3]) : 0;
  W *w = calloc((size_t)n, sizeof(W));
  pthread_t *th = calloc((size_t)n, sizeof(pthread_t));
  const double t0 = now_ms();
  for (int i = 0; i < n; ++i) {
    w[i].t0 = t0, w[i].t_end = t0 + secs * 1e3, w[i].slice_us = slice;
    pthread_create(&th[i], NULL, worker, &w[i]);
  }
  int late = 0, cut = 0, wakes = 0, rc = 0;
  double late_ms = 0, cut_ms = 0, worst = 0;
  for (int i = 0; i < n; ++i) {
    pthread_join(th[i], NULL);
    late += w[i].late, cut += w[i].cut, wakes += w[i].wakes, late_ms += w[i].late_ms, cut_ms += w[i].cut_ms;
    worst = w[i].worst > worst ? w[i].worst : worst;
    rc = w[i].set_rc ? w[i].set_rc : rc;
  }
  printf("threads %d  %.1f s  slice %s%d us (sched_setattr errno %d)  wake-ups %d: %d late > 0.5 ms (%.1f ms in all)  preempted while busy: %d (%.1f ms in all)  worst %.2f ms\n", n,
         secs, slice ? "" : "default ", slice, rc, wakes, late, late_ms, cut, cut_ms, worst);
  return 0;
}
