/*
 * herald_ps.h -- the libps.so names python/hetu binds with ctypes, exported by libherald_ps.so.
 *
 * Reference interface: /root/reference/ps-lite/src/python_binding.cc:6-151 (callers:
 * python/hetu/gpu_ops/ParameterServerCommunicate.py:68-111 `self.comm.SparsePush / SparsePull /
 * SSPushPull / Wait`, python/hetu/initializers.py:28-38 `comm.InitTensor`, executor.py save / load).
 * Same names, argument order, argument meaning and return types.  Not provided (outside the embedding
 * path): Push / DDPushPull of dense tensors with server-side optimizers, PushData / PullData,
 * preduce_get_partner.  The engine is include/herald_amd.h's ha_ps_* (csrc/ps.hip).
 */
#ifndef HERALD_PS_H_
#define HERALD_PS_H_

#include "herald_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/common/dlarray.h:61-65 */
typedef struct {
    int device_id;
    void *handle;
} DLEvent;

void Init(void);                                                             /* python_binding.cc:8-12 */
void Finalize(void);                                                         /* :14-16 */
void StartServer(void);                                                      /* :137-140 */
void InitTensor(int node_name, int ptype, int len, int width, int init_type, /* :99-106 */
                double init_a, double init_b, unsigned long long seed,
                int otype, float lrs[], int nlr);
void SparsePull(int node_name, const DLArray *index, DLArray *value);        /* :38-43 */
void SparsePush(int node_name, const DLArray *index, const DLArray *value,   /* :31-36 */
                DLEvent *evt);
void SSPushPull(int node_name, const DLArray *inindices,                     /* :54-64 */
                const DLArray *in_arr, const DLArray *outindices,
                DLArray *out_arr, DLEvent *evt);
void SDPushPull(int node_name, const DLArray *index, const DLArray *in_arr,  /* :45-52 */
                DLArray *out_arr, DLEvent *evt);
void Pull(int node_name, DLArray *arr);                                      /* :18-20 */
void Wait(int node_id);                                                      /* :83-85 */
void BarrierWorker(void);                                                    /* :91-93 */
void Clear(int node_name);                                                   /* :108-110 */
void ClearOnServer(int node_name);                                           /* :112-114 */
void SaveParam(int node_name, char *address);                                /* :116-118 */
void LoadParam(int node_name, char *address);                                /* :120-122 */
void startRecord(char *dirPath);                                             /* :124-126 */
void getLoads(void);                                                         /* :128-130 */
void ssp_init(unsigned long long key, size_t group_size, int tolerance);     /* :132-134 */
void ssp_sync(unsigned long long key, int version);                          /* :135-137 */
int rank(void);                                                              /* :142-144 */
int nrank(void);                                                             /* :146-148 */

#ifdef __cplusplus
}
#endif
#endif /* HERALD_PS_H_ */
