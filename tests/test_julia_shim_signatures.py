"""shim/CelesteMI355X.jl cannot be executed here (no Julia): what CAN be checked is that every `ccall` in it names a symbol
the header declares and passes the header's arguments -- count and type class (pointer / 32-bit / 64-bit integer / double /
size_t), position by position -- and the header's return type; and that its struct mirrors list the header's fields in order."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_class(t):
    t = t.strip()
    if "*" in t or "[" in t:
        return "ptr"
    base = re.sub(r"\b(const|unsigned|struct)\b", "", t).split()
    ty = base[0] if base else ""
    return {"int32_t": "i32", "uint32_t": "i32", "int": "i32", "int64_t": "i64", "uint64_t": "i64", "double": "f64", "float": "f32",
            "size_t": "size", "void": "void"}[ty]


def _jl_class(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Int32": "i32", "UInt32": "i32", "Cint": "i32", "Int64": "i64", "UInt64": "i64", "Float64": "f64", "Csize_t": "size",
            "Void": "void"}[t]


def _header_prototypes():
    hdr = open(os.path.join(ROOT, "include", "celeste_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_0-9]+\s*\*?)\s*(celeste_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, re.M | re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else _split_top(args)
        protos[name] = (_c_class(ret), [_c_class(re.sub(r"\b[a-zA-Z_0-9]+(\[[0-9]*\])?$", lambda mm: mm.group(1) or "", p).strip() or p)
                                         for p in params])
    return protos


def test_every_ccall_of_the_julia_shim_matches_the_header():
    protos = _header_prototypes()
    assert len(protos) == 43
    jl = open(os.path.join(ROOT, "shim", "CelesteMI355X.jl")).read()
    calls = re.findall(r"ccall\(\(:(celeste_[a-z_0-9]+),\s*libceleste\),\s*([A-Za-z{}0-9]+),\s*\(", jl)
    assert len(calls) >= 14
    seen = set()
    for m in re.finditer(r"ccall\(\(:(celeste_[a-z_0-9]+),\s*libceleste\),\s*([A-Za-z{}0-9]+),\s*\(", jl):
        name, ret = m.group(1), m.group(2)
        assert name in protos, "%s is not declared in the header" % name
        # the argument-type tuple: balanced parentheses from the end of the match
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(jl[i], 0)
            i += 1
        types = [t for t in _split_top(jl[m.end():i - 1]) if t]
        c_ret, c_args = protos[name]
        assert _jl_class(ret) == c_ret, (name, "return", ret, c_ret)
        assert [_jl_class(t) for t in types] == c_args, (name, types, c_args)
        seen.add(name)
    # the entry points a maintainer needs are all bound
    for need in ("celeste_images_create", "celeste_ctx_create_on", "celeste_elbo_eval", "celeste_elbo_eval_batch",
                 "celeste_maximize_batch", "celeste_joint_infer", "celeste_tr_solve_batch", "celeste_host_alloc", "celeste_version",
                 "celeste_group_create", "celeste_group_destroy", "celeste_group_info", "celeste_group_elbo_eval_batch",
                 "celeste_group_maximize_batch", "celeste_group_joint_infer", "celeste_group_collectives"):
        assert need in seen, need


def test_julia_struct_mirrors_list_the_header_fields_in_order():
    hdr = open(os.path.join(ROOT, "include", "celeste_mi355x.h")).read()
    jl = open(os.path.join(ROOT, "shim", "CelesteMI355X.jl")).read()
    for jname, cname in (("CImage", "celeste_image_t"), ("CPatch", "celeste_patch_t"), ("CProblem", "celeste_problem_t"),
                         ("COptimConfig", "celeste_optim_config_t"), ("CGroupInfo", "celeste_group_info_t")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        cfields = [re.sub(r"\[.*", "", d.strip().split()[-1].lstrip("*")) for d in body.split(";") if d.strip()]
        jbody = re.search(r"struct %s\b(.*?)\nend" % jname, jl, re.S).group(1)
        jbody = re.sub(r"#.*", "", jbody)
        jfields = re.findall(r"\b([a-zA-Z_0-9]+)::", jbody)
        assert jfields == cfields, (jname, jfields, cfields)


def test_the_context_table_of_the_shim_is_only_touched_under_its_lock():
    """process_source runs under Threads.@threads (ParallelRun.jl:285, :330): every use of MI355X_CONTEXTS sits between
    lock(MI355X_CONTEXTS_LOCK) and its `finally unlock`"""
    jl = open(os.path.join(ROOT, "shim", "CelesteMI355X.jl")).read()
    code = "\n".join(ln.split("#")[0] for ln in jl.splitlines())
    uses = [m.start() for m in re.finditer(r"MI355X_CONTEXTS\b(?!_LOCK)", code)]
    assert len(uses) >= 4                                  # the declaration, register, release, look-up
    for pos in uses[1:]:
        before = code[:pos]
        locks = [m.start() for m in re.finditer(r"(?<!un)lock\(MI355X_CONTEXTS_LOCK\)", before)]
        assert locks and locks[-1] > before.rfind("unlock(MI355X_CONTEXTS_LOCK)"), code[pos - 80:pos + 40]
        assert code.find("unlock(MI355X_CONTEXTS_LOCK)", pos) > 0


def test_integration_md_shows_the_shim_as_shipped():
    """INTEGRATION.md section 1 prints the reference-side binding a maintainer would add: it must be the file under shim/, not an
    older copy of it"""
    shim = open(os.path.join(ROOT, "shim", "CelesteMI355X.jl")).read().rstrip()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert shim in doc
