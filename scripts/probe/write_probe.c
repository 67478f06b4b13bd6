// Measurement aid (not product code): does writing ONE file scale with the number of writer threads on this box?
//   write_probe FILE MiB   — pwrite of disjoint ranges by 1, 2, 4, 8 threads (buffered; ext4 and xfs take the inode lock for each)
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
static int fd; static size_t total, nt; static char* buf;
static void* work(void* a){size_t t=(size_t)a; size_t lo=total*t/nt, hi=total*(t+1)/nt; for(size_t o=lo;o<hi;){size_t n=hi-o<(4u<<20)?hi-o:(4u<<20); ssize_t w=pwrite(fd,buf+(o%(64u<<20)),n,o); if(w<=0){perror("pwrite");exit(1);} o+=w;} return 0;}
int main(int c,char**v){total=(size_t)atol(v[2])<<20; buf=malloc(128u<<20); memset(buf,'x',128u<<20);
 for(nt=1;nt<=8;nt*=2){ unlink(v[1]); fd=open(v[1],O_WRONLY|O_CREAT|O_TRUNC,0644); double t=now(); pthread_t th[8]; for(size_t i=0;i<nt;i++)pthread_create(&th[i],0,work,(void*)i); for(size_t i=0;i<nt;i++)pthread_join(th[i],0); double d=now()-t; close(fd); printf("%zu threads: %.3f s %.1f GB/s\n",nt,d,total/1e9/d);} unlink(v[1]); return 0;}
