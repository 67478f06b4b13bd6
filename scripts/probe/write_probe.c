// Measurement aid (not product code): how fast can 841 MB get into ONE new file on this box?
//   write_probe FILE MiB
// (1) pwrite of disjoint ranges by 1, 2, 4, 8 threads (buffered; ext4 / xfs / overlayfs take the inode lock for each: no scaling
//     expected), (2) the file sized first (posix_fallocate), mapped MAP_SHARED and filled by 1 .. 32 threads with memcpy — page
//     faults instead of write() calls — with the munmap timed apart.
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static int fd; static size_t total, nt; static char* buf; static char* map;
static void* work(void* a) {
    size_t t = (size_t)a, lo = total * t / nt, hi = total * (t + 1) / nt;
    for (size_t o = lo; o < hi;) { size_t n = hi - o < (4u << 20) ? hi - o : (4u << 20); ssize_t w = pwrite(fd, buf + (o % (64u << 20)), n, o); if (w <= 0) { perror("pwrite"); exit(1); } o += w; }
    return 0;
}
static void* mwork(void* a) {
    size_t t = (size_t)a, lo = total * t / nt, hi = total * (t + 1) / nt;
    for (size_t o = lo; o < hi;) { size_t n = hi - o < (2u << 20) ? hi - o : (2u << 20); memcpy(map + o, buf + (o % (64u << 20)), n); o += n; }
    return 0;
}
int main(int c, char** v) {
    if (c < 3) { fprintf(stderr, "usage: write_probe FILE MiB\n"); return 2; }
    total = (size_t)atol(v[2]) << 20; buf = malloc(128u << 20); memset(buf, 'x', 128u << 20);
    pthread_t th[64];
    for (nt = 1; nt <= 8; nt *= 2) {
        unlink(v[1]); fd = open(v[1], O_WRONLY | O_CREAT | O_TRUNC, 0644);
        double t = now();
        for (size_t i = 0; i < nt; i++) pthread_create(&th[i], 0, work, (void*)i);
        for (size_t i = 0; i < nt; i++) pthread_join(th[i], 0);
        double d = now() - t; close(fd);
        printf("pwrite, %2zu threads: %.3f s %.1f GB/s\n", nt, d, total / 1e9 / d);
    }
    for (nt = 1; nt <= 32; nt *= 2) {
        unlink(v[1]); fd = open(v[1], O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t = now();
        int e = posix_fallocate(fd, 0, (off_t)total);
        if (e) { fprintf(stderr, "posix_fallocate: %s\n", strerror(e)); return 1; }
        double t_alloc = now() - t; t = now();
        map = mmap(0, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); return 1; }
        for (size_t i = 0; i < nt; i++) pthread_create(&th[i], 0, mwork, (void*)i);
        for (size_t i = 0; i < nt; i++) pthread_join(th[i], 0);
        double t_copy = now() - t; t = now();
        munmap(map, total);
        double t_unmap = now() - t; t = now();
        close(fd);
        double t_close = now() - t;
        printf("mapped, %2zu threads: fallocate %.3f + fill %.3f (%.1f GB/s) + munmap %.3f + close %.3f = %.3f s\n", nt, t_alloc, t_copy, total / 1e9 / t_copy, t_unmap, t_close,
               t_alloc + t_copy + t_unmap + t_close);
    }
    unlink(v[1]);
    return 0;
}
