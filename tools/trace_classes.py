"""Per kernel CLASS (the classes bench.py reports rooflines for) launch count and mean duration from a rocprofv3 --kernel-trace
CSV -> JSON (profiles/rNN_kernel_trace_classes.json), which bench.py reads for the ``frac_rocprof`` fields: the roofline fraction on
the profiler's clock (dispatch included) next to the one on the kernels' own wall-clock stamps.
Usage: python tools/trace_classes.py <kernel_trace.csv> <passes> <out.json> "<command>" """
import csv
import json
import sys

CLASSES = {
    "decode_attn": ["decode_attn_kernel"],
    "decode_gemm": ["dgemm_kernel", "dg3_kernel"],
    # (the 64-channel bf16 instance is the decoders' fused tail, norm_out + conv_out with 3 output channels: counted with the class
    # conv_out has always been in)
    "igemm": ["igemm_kernelIDF16b", "gemm256l_kernel", "conv3x3_kernelIDF16bLi64"],
    "conv3x3": ["conv3x3_kernelIDF16b", "conv3x3_kernel<__bf16", "conv3x3_kernel<bool"],
    "sampler": ["sample_embed_kernel"],
}


def main():
    path, passes, out, cmd = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4]
    acc = {k: [0, 0] for k in CLASSES}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for cls, pats in CLASSES.items():
            if any(p in name for p in pats):
                acc[cls][0] += 1
                acc[cls][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                break
    res = {"source": f"rocprofv3 --kernel-trace of `{cmd}` ({passes:g} passes of the step), tools/trace_classes.py", "classes": {}}
    for cls, (n, ns) in acc.items():
        if n:
            res["classes"][cls] = {"launches_per_step": n / passes, "mean_us": ns / n / 1e3, "ms_per_step": ns / passes / 1e6,
                                   "kernel_name_filter": CLASSES[cls]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["classes"], indent=1))


if __name__ == "__main__":
    main()
