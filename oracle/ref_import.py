"""Import the REAL reference hot path (build container only).  TEST INFRASTRUCTURE ONLY.

/root/reference does not exist on the GPU box; `available()` is False there and nothing in the
gpu tests, smoke() or bench.py calls `load()`.  Used by tests/golden/make_golden.py to generate
fixtures and by tests/test_reference_live.py (skipped when the reference is absent) to check
the oracle against the reference on fresh random inputs.

What is stubbed: `cv_lib` (the author's un-vendored helper library, README.md:19-28) -- only the
three import-time symbols the path touches and none of its arithmetic.
What is real: every reference .py on the path, and its C++ extension compiled unmodified from
/root/reference/cpp_extension/src into oracle/_ref/extension.so (oracle/Makefile, target ref).
"""
import importlib.machinery
import importlib.util
import json
import os
import sys
import types

REF_ROOT = os.environ.get("SCHEMANET_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO = os.path.join(_HERE, "_ref", "extension.so")
_cache = None


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "schema_inference")) and os.path.exists(_REF_SO)


def load_ext():
    """The reference's pybind module (feat_to_v_attr, feat_to_instance_v, feat_to_e,
    feat_to_instance_e).  Needs only oracle/_ref/extension.so + torch, so it also loads on the
    GPU box (used there as the `reference` CPU baseline of bench.py)."""
    import torch  # noqa: F401  (libtorch symbols)
    loader = importlib.machinery.ExtensionFileLoader("extension", _REF_SO)
    spec = importlib.util.spec_from_loader("extension", loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def load():
    """Returns a namespace with the reference modules.  Must run in a process that has NOT
    imported this repo's same-named drop-in packages (cpp_extension, discretization,
    schema_inference)."""
    global _cache
    if _cache is not None:
        return _cache
    if not available():
        raise RuntimeError("reference not available (need /root/reference and oracle/_ref)")
    for name in ("cpp_extension", "discretization", "schema_inference", "models"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(REF_ROOT):
            raise RuntimeError(f"{name} already imported from {sys.modules[name].__file__}")

    def _mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    _mod("cv_lib")
    _mod("cv_lib.utils", to_json_str=lambda o: json.dumps(o, default=str))
    _mod("cv_lib.classification")
    _mod("cv_lib.classification.models", get_model=lambda *a, **k: None, register_models=lambda d: None)

    ext = load_ext()
    sys.modules["cpp_extension.extension"] = ext  # satisfies `from .extension import ...`
    sys.path.insert(0, REF_ROOT)
    import cpp_extension  # reference shims (cpp_extension/__init__.py:20-76)
    import discretization
    import schema_inference.graph as graph
    from schema_inference.graph import utils as graph_utils
    from schema_inference.utils.ingredient_model_wrapper import IngredientModelWrapper
    from schema_inference.loss.schema_inference_loss import SchemaInferenceLoss

    ns = types.SimpleNamespace(
        ext=ext, cpp_extension=cpp_extension, discretization=discretization, graph=graph,
        graph_utils=graph_utils, IngredientModelWrapper=IngredientModelWrapper,
        SchemaInferenceLoss=SchemaInferenceLoss,
    )
    _cache = ns
    return ns
