"""
Likelihood = ErrorModel(ForwardModel(theta)).  Mirror of reference
``binf/pdf/likelihoods.py:12-174``.

Gradient (of the energy): chain rule ``J(theta) . d E / d mock`` with the
Jacobian laid out ``[n_params x n_data]`` (reference ``:148-155``; pinned by
``binf/tests/pdf/likelihoods.py:114-119``).

Dispatch: when forward and error model both advertise a native kind
(``native_spec``) and a kind registered with ``binf_amd.native`` fuses that pair,
log-prob and gradient run in one fused kernel and the ``[C x n_data]`` mock data
never reaches HBM;
otherwise the models are called as written and the contraction is the
library's generic kernel (``binf_jacobian_contract_f64``).
"""
from binf_amd.pdf import AbstractBinfPDF


class Likelihood(AbstractBinfPDF):

    def __init__(self, name, forward_model, error_model):
        super(Likelihood, self).__init__(name)
        self._forward_model = forward_model
        self._error_model = error_model
        self._inherit_variables()
        self._setup_parameters()
        self._set_original_variables()

    # -- construction --------------------------------------------------------
    def _adopt(self, model, skip=()):
        """Take over a model's original variables: still-free ones become
        variables of the likelihood, already-fixed ones are remembered as
        original variables (so their values get injected on evaluation)."""
        for v in model._original_variables:
            if v in skip:
                continue
            if v in model.parameters:
                self._original_variables.add(v)
            else:
                self._register_variable(
                    v, differentiable=v in model.differentiable_variables)
            self.update_var_param_types(**{v: model.var_param_types[v]})

    def _inherit_fwm_variables(self):
        self._adopt(self._forward_model)

    def _inherit_em_variables(self):
        self._adopt(self._error_model, skip=('mock_data',))

    def _inherit_variables(self):
        self._inherit_fwm_variables()
        self._inherit_em_variables()

    def _setup_parameters(self):
        """Own copy of every model parameter; the model's parameter follows it
        (pinned by ``binf/tests/pdf/likelihoods.py:86-99``)."""
        for component in (self._forward_model, self._error_model):
            for p in component.get_params():
                self._register(p.name)
                self[p.name] = p.__class__(p.value, p.name, self[p.name])
                p.bind_to(self[p.name])

    def _setup_fixed_variable_parameters(self):
        for model in (self._forward_model, self._error_model):
            for p in model.get_params():
                ptype = model.var_param_types[p.name]
                model[p.name] = ptype(self[p.name].value, p.name)
                model[p.name].bind_to(self[p.name])

    @property
    def forward_model(self):
        return self._forward_model

    @property
    def error_model(self):
        return self._error_model

    # -- evaluation ------------------------------------------------------------
    def _split_variables(self, variables):
        fwm = {v: variables[v] for v in variables
               if v in self.forward_model.variables}
        em = {v: variables[v] for v in variables
              if v in self.error_model.variables}
        return fwm, em

    def _native_pair(self):
        """``(forward spec, error spec)`` if both models advertise a native kind
        (``native_spec() -> (model kind, model)``) and a registered kind fuses the
        pair (``binf_amd.native``), else None."""
        from binf_amd import native
        pair = native.model_pair(self)
        return None if pair is None else pair[:2]

    def _fused(self, which, fwm_vars, em_vars):
        from binf_amd import native
        pair = native.model_pair(self)
        if pair is None:
            return None
        (_, fwm), (_, em), hooks = pair
        return hooks[which](self, fwm, em, fwm_vars, em_vars)

    def _evaluate_log_prob(self, **variables):
        fwm_vars, em_vars = self._split_variables(variables)
        out = self._fused(0, fwm_vars, em_vars)
        if out is not None:
            return out
        mock_data = self.forward_model(**fwm_vars)
        return self.error_model.log_prob(mock_data=mock_data, **em_vars)

    def _evaluate_gradient(self, **variables):
        fwm_vars, em_vars = self._split_variables(variables)
        out = self._fused(1, fwm_vars, em_vars)
        if out is not None:
            return out
        mock_data = self.forward_model(**fwm_vars)
        dfm = self.forward_model.jacobi_matrix(**fwm_vars)
        emgrad = self.error_model.gradient(mock_data=mock_data, **em_vars)
        return contract_jacobian(dfm, emgrad)

    # -- copies ------------------------------------------------------------------
    def clone(self):
        return self.__class__(self.name, self.forward_model.clone(),
                              self.error_model.clone())

    def conditional_factory(self, **fixed_vars):
        fwm = self.forward_model.clone()
        fwm.fix_variables(**fwm._get_variables_intersection(fixed_vars))
        em = self.error_model.conditional_factory(
            **self.error_model._get_variables_intersection(fixed_vars))
        return self.__class__(self.name, fwm, em)


def contract_jacobian(dfm, emgrad):
    """``dfm.dot(emgrad)`` of the reference (``:155``), batched: ``dfm`` is
    ``[n_params x n_data]`` (shared by all chains) or ``[C x n_params x n_data]``,
    ``emgrad`` is ``[n_data]`` or ``[C x n_data]``.  Device tensors go through
    ``binf_jacobian_contract_f64`` (f64 MFMA tiles for a shared Jacobian, streamed
    row products for per-chain ones; fixed summation order); numpy values -- the
    host mirror of the reference's own unit tests -- through ``ndarray.dot``.
    There is no torch fallback: CPU tensors are refused."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and (isinstance(emgrad, torch.Tensor) or isinstance(dfm, torch.Tensor)):
        from binf_amd import _native
        _native.require_device(emgrad, 'the error-model gradient')
        _native.require_device(dfm, 'the jacobi matrix')
        return _native.jacobian_contract(dfm.contiguous(), emgrad.contiguous())
    return dfm.dot(emgrad)
