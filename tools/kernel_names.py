"""Kernel names as bench.py prints them, from what rocprofv3 writes.

rocprofv3 demangles the fp32 instances ("void (anonymous namespace)::pw_gemm_glds_kernel<float, 64, ...>(...)")
but leaves the __bf16 ones mangled (its demangler does not know DF16b), so those are decoded here: the
template arguments of this engine's kernels are only float / __bf16 / int / bool literals."""
import re


def _demangle(m: str) -> str:
    g = re.match(r"_ZN12_GLOBAL__N_1(\d+)", m) or re.match(r"_Z(\d+)", m)
    if not g:
        return m
    n, pos = int(g.group(1)), g.end()
    name, rest = m[pos:pos + n], m[pos + n:]
    if not rest.startswith("I"):
        return name
    args, i = [], 1
    while i < len(rest) and rest[i] != "E":
        if rest.startswith("DF16b", i):
            args.append("__bf16"); i += 5
        elif rest[i] == "f":
            args.append("float"); i += 1
        elif rest.startswith("Li", i) or rest.startswith("Lb", i):
            j = rest.index("E", i)
            v = rest[i + 2:j]
            args.append(("true" if v == "1" else "false") if rest[i + 1] == "b" else v.replace("n", "-")); i = j + 1
        else:
            return name   # a shape this table does not know: keep the bare name
    return f"{name}<{', '.join(args)}>"


def short(kernel_name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::|^void |\(.*", "", kernel_name).strip()
    return _demangle(name) if name.startswith("_Z") else name


if __name__ == "__main__":
    assert short("_ZN12_GLOBAL__N_114pw_gemm_kernelIDF16bLi128ELi128ELi2ELi2EEEvPKT_iS3_PS1_iiiiii12GemmEpilogue") == "pw_gemm_kernel<__bf16, 128, 128, 2, 2>"
    assert short("_ZN12_GLOBAL__N_119pw_gemm_glds_kernelIDF16bLi128ELi64ELi2ELi2ELi2ELb1EEEvPKT_i") == "pw_gemm_glds_kernel<__bf16, 128, 64, 2, 2, 2, true>"
    assert short("void (anonymous namespace)::inc_kernel<float>(float const*, int)") == "inc_kernel<float>"
    print("ok")
