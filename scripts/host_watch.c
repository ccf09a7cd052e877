// host_watch.c - bench.py's step_diag leg runs this beside the steps: every 0.5 ms it reads the host's count of RUNNABLE tasks
// (/proc/loadavg, 4th field) and notes how late its own wake-up was - a thread that sleeps all the time is the first to get a CPU,
// so its lateness is what the scheduler does to everybody.  One line per sample on stdout: "<CLOCK_MONOTONIC ms> <runnable> <late ms>".
//   host_watch <seconds>                 (no GPU involved; Python's time.perf_counter() is the same clock)
#define _GNU_SOURCE
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <unistd.h>
static volatile sig_atomic_t stop_ = 0;
static void on_term(int s) { (void)s; stop_ = 1; }
int main(int argc, char **argv) {
  signal(SIGTERM, on_term); // (bench.py ends the watch with SIGTERM: leave through main so that stdout is flushed)
  signal(SIGINT, on_term);
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  const int fd = open("/proc/loadavg", O_RDONLY);
  struct timespec next;
  clock_gettime(CLOCK_MONOTONIC, &next);
  const double t_end = next.tv_sec * 1e3 + next.tv_nsec * 1e-6 + secs * 1e3;
  static char out[1 << 22];
  setvbuf(stdout, out, _IOFBF, sizeof out);
  for (;;) {
    next.tv_nsec += 500000;
    if (next.tv_nsec >= 1000000000) next.tv_nsec -= 1000000000, ++next.tv_sec;
    clock_nanosleep(CLOCK_MONOTONIC, TIMER_ABSTIME, &next, NULL);
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6, want = next.tv_sec * 1e3 + next.tv_nsec * 1e-6;
    char buf[128];
    int run = -1;
    const ssize_t n = fd >= 0 ? pread(fd, buf, sizeof buf - 1, 0) : 0;
    if (n > 0) {
      buf[n] = 0;
      float a, b, c;
      int tot;
      if (sscanf(buf, "%f %f %f %d/%d", &a, &b, &c, &run, &tot) != 5) run = -1;
    }
    printf("%.3f %d %.3f\n", t, run, t - want);
    if (t >= t_end || stop_) break;
    if (t - want > 0.5) clock_gettime(CLOCK_MONOTONIC, &next);
  }
  return 0;
}
