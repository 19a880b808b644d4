// libherald_ps.so: the names python/hetu binds from libps.so with ctypes
// (ps-lite/src/python_binding.cc:6-151), forwarded to the in-node engine of libherald_amd.so
// (csrc/ps.hip).  Signatures and return types are the reference's: these functions return void (rank /
// nrank: int); an engine error is printed once to stderr and kept in ha_last_error().
#include <stdio.h>

#include "../../include/herald_ps.h"

static void report(int rc, const char *what) {
    if (rc != 0)
        fprintf(stderr, "[herald_ps] %s failed: %s\n", what, ha_last_error());
}

extern "C" {

void Init(void) {}          // no van / postoffice to start: the exchange is set up by the host side
void Finalize(void) {}
void StartServer(void) {}   // every rank serves its own shard

void InitTensor(int node_name, int ptype, int len, int width, int init_type, double init_a, double init_b,
                unsigned long long seed, int otype, float lrs[], int nlr) {
    (void)otype, (void)lrs, (void)nlr;   // server-side optimizers are not part of the embedding path
    report(ha_ps_init_tensor(node_name, ptype, len, width, init_type, init_a, init_b, seed), "InitTensor");
}

void SparsePull(int node_name, const DLArray *index, DLArray *value) {
    report(ha_ps_sparse_pull(node_name, index, value), "SparsePull");
}

void SparsePush(int node_name, const DLArray *index, const DLArray *value, DLEvent *evt) {
    (void)evt;   // the caller's event guards a device buffer; work here is ordered on the node's stream
    report(ha_ps_sparse_push(node_name, index, value), "SparsePush");
}

void SSPushPull(int node_name, const DLArray *inindices, const DLArray *in_arr, const DLArray *outindices,
                DLArray *out_arr, DLEvent *evt) {
    (void)evt;
    report(ha_ps_sparse_push(node_name, inindices, in_arr), "SSPushPull(push)");
    report(ha_ps_sparse_pull(node_name, outindices, out_arr), "SSPushPull(pull)");
}

void SDPushPull(int node_name, const DLArray *index, const DLArray *in_arr, DLArray *out_arr, DLEvent *evt) {
    (void)evt;
    report(ha_ps_sparse_push(node_name, index, in_arr), "SDPushPull(push)");
    report(ha_ps_dense_pull(node_name, out_arr), "SDPushPull(pull)");
}

void Pull(int node_name, DLArray *arr) { report(ha_ps_dense_pull(node_name, arr), "Pull"); }

void Wait(int node_id) { report(ha_ps_wait(node_id), "Wait"); }
void BarrierWorker(void) { report(ha_ps_barrier(), "BarrierWorker"); }
void Clear(int node_name) { report(ha_ps_clear(node_name), "Clear"); }
void ClearOnServer(int node_name) { report(ha_ps_clear(node_name), "ClearOnServer"); }
void SaveParam(int node_name, char *address) { report(ha_ps_save(node_name, address), "SaveParam"); }
void LoadParam(int node_name, char *address) { report(ha_ps_load(node_name, address), "LoadParam"); }
void startRecord(char *dirPath) { (void)dirPath; }
void getLoads(void) {}
// SSP: sparse pushes / pulls are collectives over all ranks, so no rank runs ahead (hetu_ops.py)
void ssp_init(unsigned long long key, size_t group_size, int tolerance) { (void)key, (void)group_size, (void)tolerance; }
void ssp_sync(unsigned long long key, int version) { (void)key, (void)version; }
int rank(void) { return ha_ps_rank(); }
int nrank(void) { return ha_ps_nrank(); }

}  // extern "C"
