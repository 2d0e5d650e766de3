"""The C-ABI surface of libbmf.so, checked without a GPU: the library loads, exports every symbol
include/bmf.h declares, and its pure-host helpers give the known answers.  No compute calls here."""
import os
import re

import pytest

import bucket_map_amd as bma

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared():
    text = open(os.path.join(ROOT, "include", "bmf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bmf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 20
    L = bma.lib()
    for n in names:
        assert hasattr(L, n), f"libbmf.so does not export {n}"
    assert sorted(bma.SYMBOLS) == names, "python binding and header disagree"
    assert L.bmf_abi_version() == 1


def test_locator_abi_symbols():
    from bucket_map_amd import locate
    text = open(os.path.join(ROOT, "include", "bml.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(bml_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(locate.SYMBOLS), "python binding and include/bml.h disagree"
    L = locate.lib()
    for n in names:
        assert hasattr(L, n), f"libbmf.so does not export {n}"


def test_verifier_abi_symbols():
    from bucket_map_amd import verify
    text = open(os.path.join(ROOT, "include", "bmv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(bmv_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(verify.SYMBOLS), "python binding and include/bmv.h disagree"
    L = verify.lib()
    for n in names:
        assert hasattr(L, n), f"libbmf.so does not export {n}"


def test_float32_helpers_are_the_reference_derivations():
    L = bma.lib()
    assert L.bmf_fault_from_rate(15, 0.4) == 6
    assert L.bmf_fault_from_rate(20, 0.6) == 12
    assert L.bmf_threshold(0.5, 46789) == 23394
    assert L.bmf_ceil_mul_f32(0.02, 300) == 6
    p = bma.Params.from_cli(26507)
    assert (p.num_fault, p.threshold, p.min_base_quality, p.max_candidates) == (6, 13253, 300, 30)


def test_argument_errors_need_no_gpu():
    with pytest.raises(bma.BmfError) as e:
        bma.Filter(bma.Params(num_buckets=100, q=9, k=8))          # k < q (main.cpp:193-198)
    assert e.value.code == bma.BMF_ERR_ARG
    with pytest.raises(bma.BmfError) as e:
        bma.Filter(bma.Params(num_buckets=0))
    assert e.value.code == bma.BMF_ERR_ARG


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_silent_cpu_fallback():
    # Without a device the product must fail loudly, not compute on the CPU.
    with pytest.raises(bma.BmfError) as e:
        bma.Filter(bma.Params.from_cli(100))
    assert e.value.code == bma.BMF_ERR_HIP


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_no_silent_cpu_fallback_in_locator_and_verifier():
    from bucket_map_amd import locate, verify
    with pytest.raises(locate.BmlError) as e:
        locate.LocatorScan(12, 10, 4, 6, 65836)
    assert e.value.code == 2                                         # BML_ERR_HIP
    with pytest.raises(verify.BmvError) as e:
        verify.Verifier()
    assert e.value.code == 2                                         # BMV_ERR_HIP


def test_product_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", bma.LIBBMF_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "bucket-map_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "bm_oracle" not in text and "oracle_c" not in text and "_oracle.h" not in text, \
                    f"{f} references the oracle"
