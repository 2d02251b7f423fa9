// Creates a handle and a one-rank communicator on it, runs one barrier, destroys both; prints how long each step took.
// Used by scripts/gpu_comm_hang_repro.py (a second process creating a communicator on a GPU whose first process holds one).
// On ORCVIO_ERR_TIMEOUT the process reports where its threads sit (/proc/self/task/*/{comm,wchan,syscall}).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>

#include "../../include/orcvio_msckf.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void dump_threads() {
    DIR* d = opendir("/proc/self/task");
    if (!d) return;
    while (dirent* e = readdir(d)) {
        if (e->d_name[0] == '.') continue;
        std::string base = std::string("/proc/self/task/") + e->d_name + "/";
        char buf[3][256] = {{0}, {0}, {0}};
        const char* files[3] = {"comm", "wchan", "syscall"};
        for (int i = 0; i < 3; ++i) {
            FILE* f = fopen((base + files[i]).c_str(), "r");
            if (f) { if (fgets(buf[i], 255, f)) { char* nl = strchr(buf[i], '\n'); if (nl) *nl = 0; } fclose(f); }
        }
        printf("thread %s comm=%s wchan=%s syscall=%s\n", e->d_name, buf[0], buf[1], buf[2]);
    }
    closedir(d);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const double t0 = now();
    orcvio_msckf_handle* h = nullptr;
    int rc = orcvio_msckf_create(0, 8, 64, 1024, &h);
    printf("create rc %d (%.2f s)\n", rc, now() - t0);
    if (rc != 0) { printf("error: %s\n", orcvio_msckf_last_error()); return 2; }
    uint8_t id[ORCVIO_COMM_ID_BYTES];
    double t1 = now();
    rc = orcvio_msckf_comm_unique_id(id);
    printf("unique_id rc %d (%.2f s)\n", rc, now() - t1);
    if (rc != 0) { printf("error: %s\n", orcvio_msckf_last_error()); if (rc == ORCVIO_ERR_TIMEOUT) dump_threads(); return 10 + rc; }
    t1 = now();
    rc = orcvio_msckf_comm_init(h, id, 0, 1);
    printf("comm_init rc %d (%.2f s)\n", rc, now() - t1);
    if (rc != 0) { printf("error: %s\n", orcvio_msckf_last_error()); if (rc == ORCVIO_ERR_TIMEOUT) dump_threads(); return 20 + rc; }
    t1 = now();
    rc = orcvio_msckf_comm_barrier(h);
    printf("barrier rc %d (%.2f s)\n", rc, now() - t1);
    if (rc != 0) { printf("error: %s\n", orcvio_msckf_last_error()); if (rc == ORCVIO_ERR_TIMEOUT) dump_threads(); return 30 + rc; }
    orcvio_msckf_destroy(h);
    printf("probe ok (%.2f s)\n", now() - t0);
    return 0;
}
