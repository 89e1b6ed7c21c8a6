"""Name -> class registry with the reference's call surface (models/models.py:7-23 of the
reference): ``register(name)`` decorator, ``make(model_spec, args=None, load_sd=False)`` and the
module-level ``models`` dict."""
import copy

models = {}


def register(name):
    def _wrap(cls):
        models[name] = cls
        return cls
    return _wrap


def make(model_spec, args=None, load_sd=False):
    kwargs = model_spec['args']
    if args is not None:
        kwargs = copy.deepcopy(kwargs)
        kwargs.update(args)
    model = models[model_spec['name']](**kwargs)
    if load_sd:
        model.load_state_dict(model_spec['sd'])
    return model
