// herd.c - a stand-in for a co-tenant whose threads all become runnable at once every 100 ms (what a job throttled by its own CPU
// quota looks like from outside: when its period refills, every one of its threads runs at the same instant):
//   herd <threads> <burst_us> <seconds> [period_ms = 100]
// every thread sleeps until the next multiple of the period (absolute), spins `burst_us`, sleeps again.   (no GPU involved)
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
static double burst_ms, t_end, period_ms = 100.0;
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static void *run(void *arg) {
  (void)arg;
  while (now_ms() < t_end) {
    const double next = ((long long)(now_ms() / period_ms) + 1) * period_ms;
    struct timespec ts = {(time_t)(next / 1e3), (long)((next - (long long)(next / 1e3) * 1e3) * 1e6)};
    clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &ts, NULL);
    const double until = now_ms() + burst_ms;
    while (now_ms() < until) {
    }
  }
  return NULL;
}
int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128;
  burst_ms = (argc > 2 ? atof(argv[2]) : 3000) / 1e3;
  t_end = now_ms() + (argc > 3 ? atof(argv[3]) : 5) * 1e3;
  if (argc > 4) period_ms = atof(argv[4]);
  pthread_t *th = calloc((size_t)n, sizeof *th);
  for (int i = 0; i < n; ++i) pthread_create(&th[i], NULL, run, NULL);
  for (int i = 0; i < n; ++i) pthread_join(th[i], NULL);
  return 0;
}
