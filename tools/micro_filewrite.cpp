// How fast does this box take one large file?  (a) one write() stream, (b) T threads pwrite() into disjoint ranges of the same
// file (after ftruncate, and without).  usage: micro_filewrite <path> <GiB> <threads>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* path = argv[1]; const size_t gib = (size_t)atol(argv[2]); const int T = atoi(argv[3]);
    const size_t PIECE = (size_t)256 << 20, total = gib << 30;
    char* buf = (char*)malloc(PIECE); memset(buf, 7, PIECE);
    for (int mode = 0; mode < 3; ++mode) {
        unlink(path);
        const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
        const double t0 = now();
        if (mode == 2 && ftruncate(fd, (off_t)total) != 0) return 1;
        for (size_t off = 0; off < total; off += PIECE) {
            if (mode == 0) { if (write(fd, buf, PIECE) != (ssize_t)PIECE) return 2; }
            else {
                std::vector<std::thread> th;
                const size_t per = PIECE / T;
                for (int t = 0; t < T; ++t) th.emplace_back([=] { if (pwrite(fd, buf + t * per, per, (off_t)(off + t * per)) != (ssize_t)per) abort(); });
                for (auto& x : th) x.join();
            }
        }
        close(fd);
        const double dt = now() - t0;
        printf("%s: %.2f s, %.1f GB/s\n", mode == 0 ? "one write() stream" : mode == 1 ? "threads pwrite (growing file)" : "threads pwrite (after ftruncate)", dt, total / dt / 1e9);
    }
    unlink(path);
    return 0;
}
