// Can P writers fill ONE output file faster than one stream?  The multi-GPU build's last stage is a P-way merge of the ranks'
// shard files by (filter value, key) into the database file (db_merge.cpp; merge_stage2's role, ipk/src/db_builder.cpp:392-458);
// with per-rank offset lists every rank could place its own records -- if the file system lets several writers run.
//   A  one write() stream                                   (what db_merge.cpp does)
//   B  P threads, each pwrite()s its own contiguous 1/P of the file, 16 MiB per call
//   C  P threads memcpy interleaved 320-byte records (record i belongs to thread i % P) into a shared mapping of the file
// usage: micro_filewrite2 <dir> <GiB> <P>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const std::string path = std::string(argv[1]) + "/ipk_micro_filewrite.bin";
    const size_t total = (size_t)atol(argv[2]) << 30; const int P = atoi(argv[3]);
    const size_t PIECE = (size_t)16 << 20, REC = 320;
    char* buf = (char*)malloc(PIECE); memset(buf, 7, PIECE);
    for (int mode = 0; mode < 3; ++mode) {
        unlink(path.c_str());
        const int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
        if (fd < 0) { perror("open"); return 1; }
        const double t0 = now();
        if (mode == 0) {
            for (size_t off = 0; off < total; off += PIECE) if (write(fd, buf, PIECE) != (ssize_t)PIECE) return 2;
        } else if (mode == 1) {
            if (ftruncate(fd, (off_t)total) != 0) return 3;
            std::vector<std::thread> th;
            const size_t per = total / P;
            for (int t = 0; t < P; ++t) th.emplace_back([=] {
                for (size_t off = 0; off < per; off += PIECE) if (pwrite(fd, buf, PIECE, (off_t)(t * per + off)) != (ssize_t)PIECE) abort();
            });
            for (auto& x : th) x.join();
        } else {
            if (ftruncate(fd, (off_t)total) != 0) return 3;
            char* m = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) { perror("mmap"); return 4; }
            std::vector<std::thread> th;
            const size_t nrec = total / REC;
            for (int t = 0; t < P; ++t) th.emplace_back([=] { for (size_t i = t; i < nrec; i += P) memcpy(m + i * REC, buf + (i % 1000) * REC, REC); });
            for (auto& x : th) x.join();
            munmap(m, total);
        }
        close(fd);
        const double dt = now() - t0;
        printf("%-28s %s: %.2f s, %.1f GB/s\n", argv[1], mode == 0 ? "A one write() stream      " : mode == 1 ? "B threads pwrite own range" : "C threads memcpy records  ", dt, total / dt / 1e9);
        fflush(stdout);
    }
    unlink(path.c_str());
    return 0;
}
