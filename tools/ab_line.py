import json, sys
d = json.loads(sys.stdin.read())
k = d["kernels"]
print(" bench us/step %.2f | isolated fwd %.2f bwd %.2f" % (
    d["ms_per_step"] * 1e3, k["fwd_fused_kernel(gather+rank)"]["isolated_us"],
    k["bwd_fused_kernel(sgd apply+finish)"]["isolated_us"]))
