"""Marker base class for priors (reference ``binf/pdf/priors.py:10-12``)."""
from binf_amd.pdf import AbstractBinfPDF


class AbstractPrior(AbstractBinfPDF):
    pass
