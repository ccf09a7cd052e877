// stall_probe.c - who stalls a GPU box's threads every ~100 ms?  (no GPU involved)
//
//   stall_probe <spinners> <seconds> [bind_first_cpu bind_n_cpus]
//
// `spinners` threads spin on clock_gettime and log every gap > 200 us between two reads (the thread was not running);
// one SLEEPER thread sleeps 500 us at a time (clock_nanosleep, absolute) and logs every wake-up that is > 300 us late.
// A thread that sleeps most of the time has all the scheduler's credit: it preempts a competitor's CPU hog at once.
//   * sleeper late together with the spinners, at a fixed period   -> the whole cgroup was off the CPUs (bandwidth throttling
//                                                                      by a cgroup level we cannot see, or the machine itself)
//   * spinners gapped, sleeper on time                              -> the spinners lost their CPUs to somebody's runnable tasks
//   * gaps already with ONE spinner                                 -> not this process's own CPU quota
// Output: one line per event "<thread> <t_ms since start> <gap_ms>", then a summary; cgroup cpu.stat before / after.
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define MAX_EV 4096
typedef struct {
  int id;
  double t0, t_end;
  int n;
  double at[MAX_EV], gap[MAX_EV];
  double cpu_ms;
} Log;

static void *spinner(void *arg) {
  Log *lg = (Log *)arg;
  double last = now_ms();
  while (last < lg->t_end) {
    const double t = now_ms();
    if (t - last > 0.2 && lg->n < MAX_EV) lg->at[lg->n] = last - lg->t0, lg->gap[lg->n] = t - last, ++lg->n;
    last = t;
  }
  struct timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  lg->cpu_ms = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  return NULL;
}

static void *sleeper(void *arg) {
  Log *lg = (Log *)arg;
  struct timespec next;
  clock_gettime(CLOCK_MONOTONIC, &next);
  for (;;) {
    next.tv_nsec += 500000;
    if (next.tv_nsec >= 1000000000) next.tv_nsec -= 1000000000, ++next.tv_sec;
    clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, NULL);
    const double t = now_ms(), want = next.tv_sec * 1e3 + next.tv_nsec * 1e-6;
    if (t - want > 0.3 && lg->n < MAX_EV) lg->at[lg->n] = want - lg->t0, lg->gap[lg->n] = t - want, ++lg->n;
    if (t >= lg->t_end) break;
    if (t - want > 0.5) clock_gettime(CLOCK_MONOTONIC, &next); // (do not replay the missed wake-ups)
  }
  return NULL;
}

static void cat(const char *path) {
  FILE *f = fopen(path, "r");
  if (!f) return;
  char buf[512];
  printf("# %s:", path);
  while (fgets(buf, sizeof buf, f)) {
    buf[strcspn(buf, "\n")] = 0;
    printf(" %s |", buf);
  }
  printf("\n");
  fclose(f);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1;
  const double secs = argc > 2 ? atof(argv[2]) : 1.0;
  if (argc > 4) {
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c = atoi(argv[3]); c < atoi(argv[3]) + atoi(argv[4]); ++c) CPU_SET(c, &set);
    if (sched_setaffinity(0, sizeof set, &set)) perror("sched_setaffinity");
  }
  cpu_set_t set;
  sched_getaffinity(0, sizeof set, &set);
  printf("# spinners %d, %.1f s, affinity %d CPUs\n", n, secs, CPU_COUNT(&set));
  cat("/sys/fs/cgroup/cpu.max");
  cat("/sys/fs/cgroup/cpu.stat");
  cat("/sys/fs/cgroup/cpu/cpu.stat");
  cat("/proc/pressure/cpu");
  Log *logs = calloc((size_t)n + 1, sizeof(Log));
  pthread_t *th = calloc((size_t)n + 1, sizeof(pthread_t));
  const double t0 = now_ms();
  for (int i = 0; i <= n; ++i) {
    logs[i].id = i;
    logs[i].t0 = t0;
    logs[i].t_end = t0 + secs * 1e3;
    pthread_create(&th[i], NULL, i == n ? sleeper : spinner, &logs[i]);
  }
  for (int i = 0; i <= n; ++i) pthread_join(th[i], NULL);
  // events of the sleeper and of the first three spinners in full; the others as counts
  for (int i = 0; i <= n; ++i) {
    double tot = 0, mx = 0;
    for (int k = 0; k < logs[i].n; ++k) tot += logs[i].gap[k], mx = logs[i].gap[k] > mx ? logs[i].gap[k] : mx;
    printf("%s %d: %d gaps, %.1f ms in all, longest %.2f ms, cpu %.0f ms\n", i == n ? "sleeper" : "spinner", i, logs[i].n, tot, mx, logs[i].cpu_ms);
    if (i == n || i < 3)
      for (int k = 0; k < logs[i].n && k < 60; ++k)
        if (logs[i].gap[k] > 1.0) printf("   %s%d at %8.1f ms: %.2f ms\n", i == n ? "S" : "s", i, logs[i].at[k], logs[i].gap[k]);
  }
  // how many spinners were off the CPU at the same time: a histogram over 1-ms bins of the gaps > 1 ms
  {
    const int bins = (int)(secs * 1e3) + 1;
    int *off = calloc((size_t)bins, sizeof(int));
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < logs[i].n; ++k)
        if (logs[i].gap[k] > 1.0)
          for (int b = (int)logs[i].at[k]; b < (int)(logs[i].at[k] + logs[i].gap[k]) && b < bins; ++b)
            if (b >= 0) ++off[b];
    printf("# ms bins in which at least half of the spinners were off the CPU:");
    int run = 0;
    for (int b = 0; b < bins; ++b) {
      if (off[b] * 2 >= n && n > 0) {
        if (!run) printf(" [%d", b);
        run = 1;
      } else if (run) {
        printf("..%d)", b);
        run = 0;
      }
    }
    printf("\n");
  }
  cat("/sys/fs/cgroup/cpu.stat");
  cat("/sys/fs/cgroup/cpu/cpu.stat");
  cat("/proc/pressure/cpu");
  return 0;
}
