/* TEST TRANSPORT, not product: the RCCL entry points csrc/pcm_gather.cpp binds, implemented over a shared directory so that the N > 1 data path of
 * vits_pcm_gather can run with TWO processes on ONE GPU (RCCL itself refuses two ranks on one device, and this pool has one GPU per box).
 * ncclAllGather = synchronise the stream, copy the send buffer to the host, publish it as <dir>/ag_<seq>_<rank> (write + rename), wait for every peer's file
 * (bounded: 20 s, then an error — never a hang), copy the blocks into the receive buffer in rank order. Loaded through VITS_RCCL_LIB.
 * Build: gcc -shared -fPIC -O1 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/stub_rccl_shm.c -o libstub_rccl_shm.so -L/opt/rocm/lib -lamdhip64 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct { int rank, world, index; long seq; char dir[512]; } comm_t;
static int n_comms = 0; /* communicators are created in the same order on every rank: the index keeps their files apart */
static void note(const char* what, int rank, long seq) {
    const char* path = getenv("STUB_RCCL_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "%s %d %ld\n", what, rank, seq);
    fclose(f);
}
int ncclGetUniqueId(ncclUniqueId* id) {
    for (int i = 0; i < 128; ++i) id->internal[i] = (char)(i * 5 + 1 + (getpid() & 0x3f));
    return 0;
}
int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    (void)id;
    const char* d = getenv("STUB_RCCL_DIR");
    if (!d) return 1;
    comm_t* c = (comm_t*)calloc(1, sizeof(comm_t));
    c->rank = rank, c->world = nranks, c->seq = 0, c->index = n_comms++;
    snprintf(c->dir, sizeof c->dir, "%s", d);
    *comm = c;
    note("init", rank, 0);
    return 0;
}
int ncclCommDestroy(void* comm) { note("destroy", comm ? ((comm_t*)comm)->rank : -1, 0); free(comm); return 0; }
int ncclCommAbort(void* comm) { note("abort", comm ? ((comm_t*)comm)->rank : -1, 0); free(comm); return 0; }
const char* ncclGetErrorString(int r) { return r == 2 ? "stub: a peer never arrived (20 s)" : "stub error"; }
int ncclAllGather(const void* send, void* recv, size_t n, int dt, void* comm, void* stream) {
    (void)dt;
    comm_t* c = (comm_t*)comm;
    const long seq = c->seq++;
    note("allgather", c->rank, (long)n);
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
    char* host = (char*)malloc(n ? n : 1);
    if (hipMemcpy(host, send, n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    char tmp[700], fin[700];
    snprintf(tmp, sizeof tmp, "%s/ag_%d_%ld_%d.tmp", c->dir, c->index, seq, c->rank);
    snprintf(fin, sizeof fin, "%s/ag_%d_%ld_%d", c->dir, c->index, seq, c->rank);
    FILE* f = fopen(tmp, "wb");
    if (!f) return 1;
    fwrite(&n, sizeof n, 1, f);
    fwrite(host, 1, n, f);
    fclose(f);
    if (rename(tmp, fin) != 0) return 1;
    for (int r = 0; r < c->world; ++r) {
        snprintf(fin, sizeof fin, "%s/ag_%d_%ld_%d", c->dir, c->index, seq, r);
        const time_t t0 = time(NULL);
        FILE* g = NULL;
        while (!(g = fopen(fin, "rb"))) {
            if (time(NULL) - t0 > 20) { free(host); return 2; }
            usleep(2000);
        }
        size_t m = 0;
        if (fread(&m, sizeof m, 1, g) != 1 || m != n) { fclose(g); free(host); return 1; }  /* every rank must send the same count */
        if (fread(host, 1, n, g) != n) { fclose(g); free(host); return 1; }
        fclose(g);
        if (hipMemcpy((char*)recv + (size_t)r * n, host, n, hipMemcpyHostToDevice) != hipSuccess) { free(host); return 1; }
    }
    free(host);
    return 0;
}
