"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/mdd_hip.h declares,
and its host-only queries (parameter table, workspace plan) agree with the oracle.  No compute."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _lib():
    from multimodal_dataset_distillation_amd import _lib, build_ext
    if not os.path.exists(_lib.LIB_PATH):
        build_ext.build(verbose=False)
    return _lib


def test_header_symbols_all_exported():
    lib = _lib().load()
    hdr = open(os.path.join(ROOT, "include", "mdd_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mdd_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from multimodal_dataset_distillation_amd._lib import SIGNATURES
    assert declared == set(SIGNATURES), declared ^ set(SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    m = re.search(r"#define\s+MDD_ABI_VERSION\s+(\d+)", hdr)
    assert m and lib.mdd_version() == int(m.group(1)) == _lib().ABI_VERSION


@pytest.mark.parametrize("variant,expect", [("nfnet_l0", 32769488), ("nfnet_tiny", None)])
def test_param_table_matches_oracle_and_reference_counts(variant, expect):
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr, nfnet_ref as nr
    d_txt = 768 if variant == "nfnet_l0" else 32
    eng = UnrollEngine(variant, batch=10, image_size=224 if variant == "nfnet_l0" else 64,
                       d_txt=d_txt, syn_steps=2, dtype="f32", bind=False)
    fi = dr.FlatModule(nr.ImageEncoder(variant))
    ft = dr.FlatModule(dr.ProjectionHead(d_txt, eng.feature_dim))
    assert eng.feature_dim == fi.module.model.num_features
    if expect:
        assert eng.P_img == expect            # 32,769,488 (35.07 M with timm's head)
        assert eng.P_txt == 7087104           # networks.py:625-646 with 768 -> 2304
        assert eng.feature_dim == 2304        # networks.py:811
    for which, fm in (("img", fi), ("txt", ft)):
        tab = eng.param_table(which)
        assert [t[0] for t in tab] == fm.names
        assert [tuple(t[1]) for t in tab] == fm.shapes
        assert [t[2] for t in tab] == np.cumsum([0] + fm.numels[:-1]).tolist()
    assert eng.workspace_bytes > 0
    eng.close()


def test_invalid_arguments_fail_loudly():
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    with pytest.raises(RuntimeError, match="invalid argument"):
        UnrollEngine("no_such_net", bind=False)
    with pytest.raises(RuntimeError, match="invalid argument"):
        UnrollEngine("nfnet_tiny", image_size=50, bind=False)


def test_product_path_needs_gpu_no_cpu_fallback():
    import torch
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU path"):
        UnrollEngine("nfnet_tiny", batch=4, image_size=64, d_txt=32, syn_steps=1, dtype="f32")


def test_c2_workspace_fits_hbm():
    """BASELINE config 2 (N=100, syn_steps=8, bf16) must fit one MI355X (288 GB HBM3E)."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    eng = UnrollEngine("nfnet_l0", batch=100, image_size=224, d_txt=768, syn_steps=8, dtype="bf16",
                       bind=False)
    gib = eng.workspace_bytes / 2**30
    print("C2 workspace GiB:", gib)
    assert gib < 240
    eng.close()


def test_param_tables_of_every_variant_match_the_oracle_flatten_order():
    """Spec-only engines (no GPU call): the C-ABI parameter table (name, shape, offset) of every
    supported topology equals the oracle's ReparamModule-order flattening (reparam_module.py:28-39)."""
    import numpy as np
    import torch
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    from oracle import distill_ref as dr, nfnet_ref as nr
    for variant, size in (("nfnet_l0", 224), ("nfnet_l1", 224), ("nfnet_tiny", 64)):
        eng = UnrollEngine(variant, batch=4, image_size=size, d_txt=768, syn_steps=1, dtype="bf16", bind=False)
        torch.manual_seed(0)
        enc = nr.ImageEncoder(variant)
        fi = dr.FlatModule(enc)
        ft = dr.FlatModule(dr.ProjectionHead(768, enc.model.num_features))
        for tab, fm in ((eng.param_table("img"), fi), (eng.param_table("txt"), ft)):
            assert [t[0] for t in tab] == fm.names
            assert [tuple(t[1]) for t in tab] == [tuple(s) for s in fm.shapes]
            assert [t[2] for t in tab] == np.cumsum([0] + fm.numels[:-1]).tolist()
        assert eng.feature_dim == enc.model.num_features
        eng.close()
    # ViT topologies, with and without timm's classifier head (the reference's 'vit', networks.py:668, keeps it)
    from oracle import vit_ref as vr
    for variant, size in (("vit_micro", 32), ("vit_micro_cls", 32), ("vit_tiny16_cls", 224)):
        eng = UnrollEngine(variant, batch=4, image_size=size, d_txt=512, syn_steps=1, dtype="bf16", bind=False)
        fi = dr.FlatModule(vr.ImageEncoder(variant, img_size=size))
        tab = eng.param_table("img")
        assert [t[0] for t in tab] == fi.names and [tuple(t[1]) for t in tab] == [tuple(s) for s in fi.shapes]
        assert eng.feature_dim == fi.module.model.num_features
        eng.close()
    from multimodal_dataset_distillation_amd.networks import VARIANTS
    ev = UnrollEngine(VARIANTS["vit"], batch=4, image_size=224, d_txt=768, syn_steps=1, dtype="bf16", bind=False)
    assert ev.P_img == 5_717_416 and ev.feature_dim == 1000     # timm's published vit_tiny_patch16_224 size, head included
    ev.close()
    # anchors: timm's published nfnet_l0 size minus its 1000-way head
    e0 = UnrollEngine("nfnet_l0", batch=4, bind=False)
    assert e0.P_img == 32769488 and e0.P_img + 2304 * 1000 + 1000 == 35074488 and e0.P_txt == 7087104
    e0.close()


def test_c4_workspace_fits_hbm_with_the_recompute_policy():
    """BASELINE config 4 (COCO, 500 pairs, syn_steps=16, NFNet-l1, bf16; mode A = every GPU runs the
    whole unroll on its own expert): keeping all 16 steps' activations needs > 1 TB per GPU; with the
    step-level stash policy (keep_steps=0 -> one shared slot, every step but the last
    recomputed in the reverse sweep) the plan fits one MI355X's 288 GB, with keep_steps=1 as well."""
    from multimodal_dataset_distillation_amd.engine import UnrollEngine
    sizes = {}
    for keep in (None, 0, 1):
        eng = UnrollEngine("nfnet_l1", batch=500, num_queries=500, image_size=224, d_txt=768, syn_steps=16,
                           dtype="bf16", bind=False, keep_steps=keep)
        sizes[keep] = eng.workspace_bytes
        assert eng.num_slots == (16 if keep is None else keep + 1)
        eng.close()
    print("C4 workspace GiB:", {k: round(v / 2**30, 1) for k, v in sizes.items()})
    assert sizes[None] > 1000e9                      # not runnable without the policy
    expert_pair = 2 * 4 * (60_228_712 + 9_449_472)   # theta0 + theta* of one expert (l1 image encoder + 768->3072 head), fp32
    for keep in (0, 1):
        assert sizes[keep] + 2 * 500 * (3 * 224 * 224 + 768) * 4 + expert_pair < 288e9
    # same policy on the benched config: 8 kept steps -> 1 shared slot
    e = UnrollEngine("nfnet_l0", batch=100, image_size=224, d_txt=768, syn_steps=8, dtype="bf16", bind=False,
                     keep_steps=0)
    assert e.workspace_bytes < 0.3 * 79.9 * 2**30 * 1.0 + 2**30
    e.close()
