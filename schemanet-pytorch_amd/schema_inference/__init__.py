"""`schema_inference` -- hot-path subset (graph/, utils/ingredient_model_wrapper) of the
reference package plus the training-step closure (loss/, train.py), MI355X-native.

`__path__` is extended so that, when the reference checkout is also on sys.path (after this
directory), its orchestration sub-packages that are NOT re-implemented here
(schema_inference.tasks / eval / data) still resolve, while `schema_inference.graph`,
`schema_inference.loss` and `schema_inference.utils` resolve to this implementation.  See INTEGRATION.md.
"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
