"""The JNI shim (jni/dpf_jni.c) and its Scala declarations (jni/scala/mclab/gpu/Dpf.scala) against include/dpf.h,
without a JDK: the dpf_config field list both sides hard-wire must be the header's, every native method must exist
on both sides with the same number of arguments, the shim may only call functions the header declares, and the C
file must pass a syntax-only compile against a stub of the JNI interface (tests/jni_stub/jni.h)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def _strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def header_config_fields():
    h = _strip_c_comments(_read("include", "dpf.h"))
    body = re.search(r"typedef struct dpf_config \{(.*?)\} dpf_config;", h, flags=re.S).group(1)
    fields = re.findall(r"\b(int32_t|int64_t)\s+(\w+)\s*;", body)
    return [n for t, n in fields if t == "int32_t"], [n for t, n in fields if t == "int64_t"]


def test_shim_assigns_every_config_field_in_header_order():
    ints, longs = header_config_fields()
    assert len(ints) == 20 and longs == ["row_begin", "row_end"]
    c = _strip_c_comments(_read("jni", "dpf_jni.c"))
    assert int(re.search(r"#define DPF_JNI_CONFIG_INTS (\d+)", c).group(1)) == len(ints)
    assigned = re.findall(r"cfg\.(\w+) = \(int32_t\)v\[(\d+)\];", c)
    assert [n for n, _ in assigned] == ints
    assert [int(i) for _, i in assigned] == list(range(len(ints)))
    for n in longs:
        assert re.search(r"cfg\.%s = \(int64_t\)\w+;" % n, c), n
    # a struct the shim leaves no byte of uninitialised: memset before the assignments
    assert "memset(&cfg, 0, sizeof cfg);" in c


def test_scala_field_list_is_the_headers():
    ints, _ = header_config_fields()
    s = _read("jni", "scala", "mclab", "gpu", "Dpf.scala")
    block = re.search(r"val ConfigIntFields: Array\[String\] = Array\((.*?)\)", s, flags=re.S).group(1)
    assert re.findall(r'"(\w+)"', block) == ints
    # the replaced DensevectorRDFInit names only fields that exist
    d = _read("jni", "scala", "mclab", "deploy", "DensevectorRDFInit.scala")
    used = re.findall(r'^\s*"(\w+)" -> ', d, flags=re.M)
    assert used and set(used) <= set(ints)
    for must in ("family", "dim", "num_functions", "num_tables", "chain_length", "partition_bits", "bucket_bits",
                 "dir_node_size", "bucket_overflow", "type_of_index", "metric", "top_k", "dtype"):
        assert must in used, must


def test_native_methods_match_on_both_sides():
    c = _strip_c_comments(_read("jni", "dpf_jni.c"))
    s = _read("jni", "scala", "mclab", "gpu", "Dpf.scala")
    c_fns = {}
    for m in re.finditer(r"JNICALL Java_mclab_gpu_Dpf_(\w+)\((.*?)\)\s*\{", c, flags=re.S):
        c_fns[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()]) - 2      # JNIEnv *, jclass
    s_fns = {}
    for m in re.finditer(r"@native def (\w+)\((.*?)\)\s*:", s):
        s_fns[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()])
    assert c_fns == s_fns
    assert {"create", "destroy", "setHashFamily", "setPartitioner", "setSamplingIndex", "fit", "queryBatch", "queryTopK"} <= set(c_fns)
    # everything the replaced Scala object calls exists
    d = _read("jni", "scala", "mclab", "deploy", "DensevectorRDFInit.scala")
    for name in set(re.findall(r"\bDpf\.([a-z]\w+)\(", d)):
        assert name in s_fns or name == "cfgInts", name


def test_shim_calls_only_declared_abi_functions():
    h = _strip_c_comments(_read("include", "dpf.h"))
    declared = set(re.findall(r"\b(dpf_[a-z_0-9]+)\s*\(", h))
    c = _strip_c_comments(_read("jni", "dpf_jni.c"))
    called = set(re.findall(r"\b(dpf_[a-z_0-9]+)\s*\(", c))
    assert called and called <= declared, called - declared
    # the reference-facing entry points are all reachable from the JVM
    for must in ("dpf_create", "dpf_destroy", "dpf_set_hash_family", "dpf_set_partitioner", "dpf_set_sampling_index", "dpf_fit",
                 "dpf_query_batch_view", "dpf_query_batch_resident", "dpf_sets_fetch", "dpf_sets_topk", "dpf_query_topk",
                 "dpf_last_error", "dpf_subindex_distribution"):
        assert must in called, must


def test_shim_compiles_against_the_jni_interface_stub():
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        pytest.skip("no C compiler")
    res = subprocess.run([cc, "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                          "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "jni", "dpf_jni.c")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert res.returncode == 0, res.stdout.decode()


def _header_enum(prefix):
    h = _strip_c_comments(_read("include", "dpf.h"))
    return {n: int(v) for n, v in re.findall(r"\b(%s\w+)\s*=\s*(\d+)" % prefix, h)}


def _camel(name):
    return "".join(w.capitalize() for w in name.lower().split("_"))


def test_option_and_flag_numbers_are_the_headers_in_every_binding():
    # dpf_set_option's numbers and dpf_config.flags bits: include/dpf.h, the ctypes binding, the Scala object
    opts = _header_enum("DPF_OPT_")
    flags = _header_enum("DPF_FLAG_")
    assert len(opts) >= 12 and len(flags) >= 6
    from similaritysearchbyrdf_amd import _lib
    assert _lib.OPTION == {n[len("DPF_OPT_"):].lower(): v for n, v in opts.items()}
    s = _read("jni", "scala", "mclab", "gpu", "Dpf.scala")
    scala = {n: int(v) for n, v in re.findall(r"\bval ((?:Opt|Flag)\w+) = (\d+)", s)}
    want = {"Opt" + _camel(n[len("DPF_OPT_"):]): v for n, v in opts.items()}
    want.update({"Flag" + _camel(n[len("DPF_FLAG_"):]): v for n, v in flags.items()})
    assert scala == want
