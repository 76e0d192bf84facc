// Developer tool (GPU box): what does the box's file system take?  N threads pwrite a file of SIZE GiB in 8 MiB blocks from
// memory, plain / after fallocate / with O_DIRECT - the floor under the file driver's write side (phyNGSC.cpp:875, 925).
//   gcc -O2 -pthread -o tools/ubench_write tools/ubench_write.c && tools/ubench_write /tmp/ubw.bin 12 8
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static const size_t BLK = 8u << 20;
static int g_fd;
static uint64_t g_blocks;
static int g_threads;
static uint8_t *g_buf;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void *work(void *arg)
{
    const long id = (long)arg;
    for (uint64_t b = id; b < g_blocks; b += g_threads) {
        size_t done = 0;
        while (done < BLK) {
            ssize_t k = pwrite(g_fd, g_buf + done, BLK - done, (off_t)(b * BLK + done));
            if (k <= 0) { perror("pwrite"); return (void *)1; }
            done += (size_t)k;
        }
    }
    return NULL;
}

static void run(const char *path, const char *label, int flags, int prealloc)
{
    unlink(path);
    g_fd = open(path, O_WRONLY | O_CREAT | flags, 0644);
    if (g_fd < 0) { printf("%-28s open failed (%s)\n", label, flags & O_DIRECT ? "O_DIRECT not supported here" : "?"); return; }
    double t0 = now();
    if (prealloc && posix_fallocate(g_fd, 0, (off_t)(g_blocks * BLK)) != 0) { printf("%-28s fallocate failed\n", label); close(g_fd); return; }
    double t1 = now();
    pthread_t th[64];
    for (long i = 0; i < g_threads; ++i) pthread_create(&th[i], NULL, work, (void *)i);
    long bad = 0;
    for (int i = 0; i < g_threads; ++i) { void *r; pthread_join(th[i], &r); bad |= (long)r; }
    double t2 = now();
    close(g_fd);
    double t3 = now();
    printf("%-28s %s prealloc %.2f s, write %.2f s = %.2f GB/s, close %.2f s\n", label, bad ? "FAILED" : "ok", t1 - t0, t2 - t1,
           g_blocks * (double)BLK / (t2 - t1) / 1e9, t3 - t2);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s <file> <GiB> <threads>\n", argv[0]); return 1; }
    g_blocks = (uint64_t)atoll(argv[2]) * 128;
    g_threads = atoi(argv[3]);
    if (posix_memalign((void **)&g_buf, 1 << 21, BLK)) return 1;
    memset(g_buf, 0x5a, BLK);
    run(argv[1], "page cache", 0, 0);
    run(argv[1], "page cache + fallocate", 0, 1);
    run(argv[1], "O_DIRECT", O_DIRECT, 0);
    run(argv[1], "O_DIRECT + fallocate", O_DIRECT, 1);
    unlink(argv[1]);
    return 0;
}
