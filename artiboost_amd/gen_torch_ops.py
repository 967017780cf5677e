"""Generates csrc/torch_ops_gen.cpp: every entry point of the C ABI (include/artiboost_hip.h) registered with the PyTorch
dispatcher as `torch.ops.artiboost_hip.<name>` (the `ab_` prefix dropped) -- the "PyTorch-ROCm custom ops" form of the boundary
that BASELINE.json's north_star / SURVEY.md section 8b name.

Mapping (mechanical, from the C declaration):
  * any pointer parameter          -> `Tensor? name`  (None = NULL).  Device pointers are device tensors; HOST pointers (the
    `ab_scene` / `ab_symcorner` / descriptor structs, host weight arrays) are CPU byte / float tensors over the same bytes.
    Non-const pointers are declared mutable: `Tensor(a!)? name`.
  * int / long / float              -> int / int / float
  * the trailing `void* stream`     -> dropped: the op launches on torch's CURRENT HIP stream of the current device
  * `int` status return             -> the op returns nothing and raises (TORCH_CHECK) on a non-zero status -- no fallback path
  * `long` / `int` query functions (no pointer arguments, no stream) -> `-> int`
The ops are registered for every backend key at once (CompositeExplicitAutograd): their tensor arguments are raw buffers of mixed
devices, the kernels never look at strides.

Argument checks (round 4; SURVEY 8b "validate with TORCH_CHECK"): every wrapper checks its tensors BEFORE the C call -- device / host placement,
contiguity, the dtype a typed pointer implies, and the `@check` clauses of the header's "Argument contracts" block (element / byte counts as
expressions of the op's integer arguments, dtypes of `void*` arguments).  A violation raises RuntimeError; nothing is launched.

Functional forms (`torch.ops.artiboost_hip.<op>.fn`): tensor-in / tensor-out overloads that derive every size from the tensors' shapes and
ALLOCATE their outputs, for the ops a maintainer of the reference would bind directly -- mano_lbs, conv2d_fwd_x3, render_batch (FUNCTIONAL)."""
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "artiboost_hip.h")
OUT = os.path.join(HERE, "csrc", "torch_ops_gen.cpp")


def declarations():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//.*", "", txt)
    out = []
    for rt, name, args in re.findall(r"\b(int|long|void)\s+(ab_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        params = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            m = re.match(r"(.*?)(\w+)$", a)
            params.append((m.group(1).strip(), m.group(2)))
        out.append((rt, name, params))
    return out


TYPED = {"float": "at::kFloat", "int32_t": "at::kInt", "int": "at::kInt", "int64_t": "at::kLong", "long": "at::kLong", "uint8_t": "at::kByte"}
NAMED = {"bf16": "at::kBFloat16", "u8": "at::kByte", "i32": "at::kInt", "i64": "at::kLong", "f32": "at::kFloat"}


def contracts():
    """{op name: [clause, ...]} from the `@check` lines of the header."""
    out = {}
    for name, body in re.findall(r"@check\s+(ab_\w+):\s*(.*)", open(HEADER).read()):
        out[name] = [c.strip() for c in body.strip().rstrip("*/").split(";") if c.strip()]
    return out


def _placement(ty, pn):
    """'host' | 'dev' | None (a struct table whose placement the name does not tell)."""
    if pn.endswith("_host"):
        return "host"
    if pn.endswith("_dev"):
        return "dev"
    if "ab_" in ty:
        return None
    return "dev"


def checks_for(name, ps, clauses):
    """C++ statements that validate the tensor arguments of one op."""
    op = name[3:]
    ptrs = {pn: ty for ty, pn in ps if ty.endswith("*")}
    ints = {pn for ty, pn in ps if not ty.endswith("*")}
    lines = []
    strided = set()
    for cl in clauses:                          # `strided: names` -- arguments addressed through a row-pitch argument: views are legitimate
        m = re.match(r"^strided:\s*(.+)$", cl)
        if m:
            strided |= set(m.group(1).split())
    assert strided <= set(ptrs), f"{name}: @check strided names {sorted(strided - set(ptrs))}, not pointer arguments"
    for pn, ty in ptrs.items():
        base = ty.replace("const", "").replace("*", "").strip()
        dt = TYPED.get(base, "-1")
        where = _placement(ty, pn)
        if where == "host":
            lines.append(f'ck_host(OP, "{pn}", {pn});')
        elif where == "dev":
            lines.append(f'ck_dev(OP, "{pn}", {pn}, (int){dt}, {"false" if pn in strided else "true"});')
        elif pn not in strided:
            lines.append(f'ck_contig(OP, "{pn}", {pn});')
    for cl in clauses:
        if cl.startswith("strided:"):
            continue
        m = re.match(r"^dt\((\w+)\):\s*(.+)$", cl)
        if m:
            assert m.group(1) in ints, f"{name}: @check dt({m.group(1)}) is not an integer argument"
            for pn in m.group(2).split():
                assert pn in ptrs, f"{name}: @check names {pn!r}, not a pointer argument"
                lines.append(f'ck_dtcode(OP, "{pn}", {pn}, {m.group(1)});')
            continue
        m = re.match(r"^(bf16|u8|i32|i64|f32):\s*(.+)$", cl)
        if m:
            for pn in m.group(2).split():
                assert pn in ptrs, f"{name}: @check names {pn!r}, not a pointer argument"
                lines.append(f'ck_dtype(OP, "{pn}", {pn}, {NAMED[m.group(1)]});')
            continue
        m = re.match(r"^(bytes\s+)?([\w\s]+?)\s*>=\s*(.+)$", cl)
        assert m, f"{name}: cannot read @check clause {cl!r}"
        fn = "ck_bytes" if m.group(1) else "ck_numel"
        for pn in m.group(2).split():
            assert pn in ptrs, f"{name}: @check names {pn!r}, not a pointer argument"
            lines.append(f'{fn}(OP, "{pn}", {pn}, (int64_t)({m.group(3)}));')
    return lines


def _alias(i):
    return "abcdefghijklmnopqrstuvwxyz"[i] if i < 26 else "a" + "abcdefghijklmnopqrstuvwxyz"[i - 26]


def schema_and_wrapper(rt, name, params, clauses=()):
    op = name[3:]
    has_stream = bool(params) and params[-1] == ("void*", "stream")
    ps = params[:-1] if has_stream else params
    sch, cargs, cparams, nmut = [], [], [], 0
    for ty, pn in ps:
        if ty.endswith("*"):
            const = ty.startswith("const")
            if const:
                sch.append(f"Tensor? {pn}")
            else:
                sch.append(f"Tensor({_alias(nmut)}!)? {pn}")
                nmut += 1
            cparams.append(f"const c10::optional<at::Tensor>& {pn}")
            cargs.append(f"({ty})p({pn})")
        elif ty in ("int", "long"):
            sch.append(f"int {pn}")
            cparams.append(f"int64_t {pn}")
            cargs.append(f"({ty}){pn}")
        elif ty == "float":
            sch.append(f"float {pn}")
            cparams.append(f"double {pn}")
            cargs.append(f"(float){pn}")
        else:
            raise ValueError(f"{name}: unhandled parameter type {ty!r}")
    if has_stream:
        cargs.append("cur_stream()")
    query = not has_stream and not any(t.endswith("*") for t, _ in ps)
    call = f"{name}({', '.join(cargs)})"
    if query:
        schema = f"{op}({', '.join(sch)}) -> int"
        body = f"static int64_t w_{op}({', '.join(cparams)}) {{ return (int64_t){call}; }}"
    else:
        schema = f"{op}({', '.join(sch)}) -> ()"
        chk = checks_for(name, ps, clauses)
        pre = f'static const char* const OP = "artiboost_hip::{op}"; ' + " ".join(chk) + " " if chk else ""
        if rt == "void":
            body = f"static void w_{op}({', '.join(cparams)}) {{ {pre}{call}; }}"
        else:
            body = (f"static void w_{op}({', '.join(cparams)}) {{ {pre}const long rc = (long){call}; "
                    f"TORCH_CHECK(rc == 0, \"{name} failed with code \", rc, rc > 0 ? \" (hipError)\" : \" (argument error)\"); }}")
    return op, schema, body


CHECK_HELPERS = r"""
// ---- argument checks (see the "Argument contracts" block of the header)
#include <c10/hip/HIPFunctions.h>
typedef c10::optional<at::Tensor> OptT;
static inline bool has(const OptT& t) { return t.has_value() && t->defined(); }
static inline void ck_contig(const char* op, const char* a, const OptT& t) {
    if (has(t)) TORCH_CHECK(t->is_contiguous(), op, ": argument '", a, "' must be contiguous (the kernels take raw buffers)");
}
static inline void ck_dev(const char* op, const char* a, const OptT& t, int dt, bool contiguous) {
    if (!has(t)) return;
    TORCH_CHECK(t->is_cuda(), op, ": argument '", a, "' must be a HIP tensor, got a ", t->device().str(), " tensor");
    TORCH_CHECK(t->get_device() == c10::hip::current_device(), op, ": argument '", a, "' lives on device ", t->get_device(), ", the current device is ", (int)c10::hip::current_device());
    if (contiguous) TORCH_CHECK(t->is_contiguous(), op, ": argument '", a, "' must be contiguous (the kernel takes it as a raw buffer without a pitch)");
    if (dt >= 0) TORCH_CHECK((int)t->scalar_type() == dt, op, ": argument '", a, "' must be ", c10::toString((at::ScalarType)dt), ", got ", c10::toString(t->scalar_type()));
}
static inline void ck_host(const char* op, const char* a, const OptT& t) {
    if (!has(t)) return;
    TORCH_CHECK(t->is_cpu(), op, ": argument '", a, "' is a HOST structure / array of the C ABI and must be a CPU tensor, got a ", t->device().str(), " tensor");
    TORCH_CHECK(t->is_contiguous(), op, ": argument '", a, "' must be contiguous");
}
static inline void ck_dtype(const char* op, const char* a, const OptT& t, at::ScalarType want) {
    if (has(t)) TORCH_CHECK(t->scalar_type() == want, op, ": argument '", a, "' must be ", c10::toString(want), ", got ", c10::toString(t->scalar_type()));
}
static inline void ck_dtcode(const char* op, const char* a, const OptT& t, int64_t code) {
    // AB_DT_U8N (the loaders' integer image plane 2 v - 255) is a bfloat16 tensor as well; ops that do not write it refuse the code themselves (AB_EINVAL)
    TORCH_CHECK(code == AB_DT_F32 || code == AB_DT_BF16 || code == AB_DT_U8N, op, ": dtype code ", code, " is none of AB_DT_F32, AB_DT_BF16, AB_DT_U8N");
    ck_dtype(op, a, t, code == AB_DT_F32 ? at::kFloat : at::kBFloat16);
}
static inline void ck_numel(const char* op, const char* a, const OptT& t, int64_t need) {
    if (has(t)) TORCH_CHECK(need >= 0 && t->numel() >= need, op, ": argument '", a, "' holds ", t->numel(), " elements, the integer arguments ask for ", need);
}
static inline void ck_bytes(const char* op, const char* a, const OptT& t, int64_t need) {
    if (has(t)) TORCH_CHECK(need >= 0 && (int64_t)t->nbytes() >= need, op, ": argument '", a, "' holds ", (int64_t)t->nbytes(), " bytes, the integer arguments ask for ", need);
}
static inline int64_t co(int64_t size, int64_t k, int64_t stride, int64_t pad) { return (size + 2 * pad - k) / stride + 1; }
"""

FUNCTIONAL = r"""
// ---- functional forms: sizes from the tensors, outputs allocated here (torch.ops.artiboost_hip.<op>.fn)
static void need(bool ok, const char* op, const char* what) { TORCH_CHECK(ok, op, ": ", what); }
static std::tuple<at::Tensor, at::Tensor, at::Tensor> f_mano_lbs(const at::Tensor& pose, const at::Tensor& betas, const at::Tensor& v_template,
        const at::Tensor& shapedirs, const at::Tensor& posedirs, const at::Tensor& J_regressor, const at::Tensor& weights, const at::Tensor& hands_mean) {
    const char* OP = "artiboost_hip::mano_lbs.fn";
    need(pose.dim() == 2 && pose.size(1) == 48, OP, "pose must be [B, 48]");
    const int64_t B = pose.size(0);
    need(betas.dim() == 2 && betas.size(0) == B && betas.size(1) == 10, OP, "betas must be [B, 10]");
    need(v_template.numel() == 778 * 3 && shapedirs.numel() == 778 * 3 * 10 && posedirs.numel() == 778 * 3 * 135 && J_regressor.numel() == 16 * 778 &&
         weights.numel() == 778 * 16 && hands_mean.numel() == 45, OP, "MANO tables: v_template [778,3], shapedirs [778,3,10], posedirs [778,3,135], J_regressor [16,778], weights [778,16], hands_mean [45]");
    auto o = pose.options().dtype(at::kFloat);
    at::Tensor verts = at::empty({B, 778, 3}, o), joints = at::empty({B, 21, 3}, o), T = at::empty({B, 16, 4, 4}, o);
    w_mano_lbs(pose, betas, v_template, shapedirs, posedirs, J_regressor, weights, hands_mean, B, verts, joints, T);
    return {verts, joints, T};
}
static std::tuple<at::Tensor, at::Tensor> f_conv2d_fwd_x3(const at::Tensor& x, const at::Tensor& w, int64_t stride, int64_t pad, const OptT& bias,
                                                           bool want_stats, bool relu) {
    const char* OP = "artiboost_hip::conv2d_fwd_x3.fn";
    need(x.dim() == 5 && x.size(0) == 2 && x.scalar_type() == at::kBFloat16, OP, "x must be the (hi, lo) bf16 planes [2, N, H, W, Cin]");
    need(w.dim() == 5 && w.size(0) == 2 && w.scalar_type() == at::kBFloat16 && w.size(4) == x.size(4), OP, "w must be the (hi, lo) bf16 planes [2, Cout, kh, kw, Cin]");
    need(stride >= 1 && pad >= 0, OP, "stride >= 1, pad >= 0");
    const int64_t N = x.size(1), H = x.size(2), W = x.size(3), Cin = x.size(4), Cout = w.size(1), kh = w.size(2), kw = w.size(3);
    const int64_t Ho = co(H, kh, stride, pad), Wo = co(W, kw, stride, pad);
    need(Ho > 0 && Wo > 0, OP, "the kernel does not fit the padded input");
    at::Tensor y = at::empty({N, Ho, Wo, Cout}, x.options().dtype(at::kFloat)), stats;
    if (want_stats) stats = at::empty({(int64_t)ab_conv2d_x3_stat_rows((int)N, (int)H, (int)W, (int)Cin, (int)Cout, (int)kh, (int)kw, (int)stride, (int)pad), Cout, 2}, y.options());
    w_conv2d_fwd_x3(x[0], x[1], w[0], w[1], y, N, H, W, Cin, Cout, kh, kw, stride, pad, bias, want_stats ? OptT(stats) : OptT(), relu ? 1 : 0);
    return {y, want_stats ? stats : at::empty({0}, y.options())};
}
static std::tuple<at::Tensor, at::Tensor> f_render_batch(const at::Tensor& scene_host, const at::Tensor& samples, const at::Tensor& hand_verts,
        const at::Tensor& order, const at::Tensor& factor, const at::Tensor& inv_affine, const OptT& blur_radius, int64_t max_faces, int64_t ow,
        int64_t oh, int64_t out_dtype, bool want_chw) {
    const char* OP = "artiboost_hip::render_batch.fn";
    need(scene_host.is_cpu() && scene_host.is_contiguous() && (size_t)scene_host.nbytes() >= sizeof(ab_scene), OP, "scene_host must be the bytes of an ab_scene (CPU tensor)");
    need(hand_verts.dim() == 3 && hand_verts.size(1) == 778 && hand_verts.size(2) == 3, OP, "hand_verts must be [B, 778, 3]");
    const int64_t B = hand_verts.size(0);
    need(order.numel() == B * 4 && factor.numel() == B * 4 && inv_affine.numel() == B * 6, OP, "order / factor [B, 4], inv_affine [B, 6]");
    need(ow > 0 && oh > 0 && max_faces > 0, OP, "ow, oh, max_faces > 0");
    const ab_scene* sc = (const ab_scene*)scene_host.data_ptr();
    auto ob = hand_verts.options();
    at::Tensor pad = at::zeros({B, oh + 6, ow + 8, 4}, ob.dtype(out_dtype == AB_DT_F32 ? at::kFloat : at::kBFloat16));
    at::Tensor chw = want_chw ? at::empty({B, 3, oh, ow}, ob.dtype(at::kFloat)) : at::Tensor();
    at::Tensor ws = at::empty({(int64_t)ab_render_workspace_bytes((int)B, sc->W, sc->H, (int)max_faces)}, ob.dtype(at::kByte));
    w_render_batch(scene_host, samples, hand_verts, order, factor, inv_affine, blur_radius, B, max_faces, ow, oh, out_dtype, pad,
                   want_chw ? OptT(chw) : OptT(), ws, OptT(), OptT());
    return {pad, want_chw ? chw : at::empty({0}, ob.dtype(at::kFloat))};
}
"""
FUNCTIONAL_DEFS = [
    '    m.def("mano_lbs.fn(Tensor pose, Tensor betas, Tensor v_template, Tensor shapedirs, Tensor posedirs, Tensor J_regressor, Tensor weights, '
    'Tensor hands_mean) -> (Tensor verts, Tensor joints, Tensor T_abs)");',
    '    m.def("conv2d_fwd_x3.fn(Tensor x, Tensor w, int stride, int pad, Tensor? bias=None, bool want_stats=False, bool relu=False) -> (Tensor y, Tensor stats)");',
    '    m.def("render_batch.fn(Tensor scene_host, Tensor samples, Tensor hand_verts, Tensor order, Tensor factor, Tensor inv_affine, Tensor? blur_radius, '
    'int max_faces, int ow, int oh, int out_dtype, bool want_chw=True) -> (Tensor out_pad, Tensor out_chw)");',
]
FUNCTIONAL_IMPLS = ['    m.impl("mano_lbs.fn", &f_mano_lbs);', '    m.impl("conv2d_fwd_x3.fn", &f_conv2d_fwd_x3);', '    m.impl("render_batch.fn", &f_render_batch);']


def generate():
    decls = declarations()
    lines = ["// GENERATED by artiboost_amd/gen_torch_ops.py from include/artiboost_hip.h -- do not edit.",
             "// torch.ops.artiboost_hip.<name>: every C-ABI entry point of libartiboost_hip.so behind the PyTorch dispatcher.",
             "#include <ATen/ATen.h>", "#include <c10/hip/HIPStream.h>", "#include <torch/library.h>",
             '#include "../../include/artiboost_hip.h"', "",
             "static inline void* p(const c10::optional<at::Tensor>& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }",
             "static inline void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }", CHECK_HELPERS, ""]
    regs, impls = [], []
    cons = contracts()
    unknown = set(cons) - {d[1] for d in decls}
    assert not unknown, f"@check lines for undeclared functions: {sorted(unknown)}"
    for rt, name, params in decls:
        op, schema, body = schema_and_wrapper(rt, name, params, cons.get(name, ()))
        lines.append(body)
        regs.append(f'    m.def("{schema}");')
        impls.append(f'    m.impl("{op}", &w_{op});')
    lines += ["", FUNCTIONAL, "", "TORCH_LIBRARY(artiboost_hip, m) {"] + regs + FUNCTIONAL_DEFS + ["}", "",
              "TORCH_LIBRARY_IMPL(artiboost_hip, CompositeExplicitAutograd, m) {"] + impls + FUNCTIONAL_IMPLS + ["}", ""]
    src = "\n".join(lines)
    if not os.path.exists(OUT) or open(OUT).read() != src:
        with open(OUT, "w") as f:
            f.write(src)
    return OUT, [d[1] for d in decls]


if __name__ == "__main__":
    print(generate()[0])
