"""
Error models: densities over the ``mock_data`` variable (and nuisance
variables such as a precision).  Mirror of reference
``binf/model/errormodels.py:15-17``.  ``gradient`` is taken w.r.t.
``mock_data`` (reference ``binf/example/likelihood.py:59-61``).
"""
from binf_amd.pdf import AbstractBinfPDF


class AbstractErrorModel(AbstractBinfPDF):

    def native_spec(self):
        """Descriptor of a HIP implementation of this model, or None."""
        return None
