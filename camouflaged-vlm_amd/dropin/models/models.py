"""Registry of the drop-in model classes.

Public surface, as the reference's ``models/models.py`` offers it to ``models.make(config['model'])`` callers:
  * ``models``            -- dict: registry name -> class
  * ``register(name)``    -- class decorator adding an entry
  * ``make(spec, args=None, load_sd=False)`` -- build ``spec['name']`` with ``spec['args']`` (optionally overridden by
    ``args``) and, on request, load ``spec['sd']`` into it.
Unknown names fail with the list of registered ones instead of a bare KeyError.
"""
from typing import Any, Callable, Dict, Mapping, Optional, Type

models: Dict[str, Type[Any]] = {}


def register(name: str) -> Callable[[Type[Any]], Type[Any]]:
    """``@register('sam')`` puts the decorated class into ``models`` under that name (last definition wins)."""
    def add(cls: Type[Any]) -> Type[Any]:
        models.update({name: cls})
        return cls
    return add


def _constructor_kwargs(spec: Mapping[str, Any], overrides: Optional[Mapping[str, Any]]) -> Dict[str, Any]:
    merged = dict(spec.get('args') or {})            # shallow copy: the YAML dict of the caller stays untouched
    if overrides:
        merged.update(overrides)
    return merged


def make(model_spec: Mapping[str, Any], args: Optional[Mapping[str, Any]] = None, load_sd: bool = False):
    key = model_spec['name']
    try:
        cls = models[key]
    except KeyError:
        raise KeyError(f"model {key!r} is not registered (known: {sorted(models)})") from None
    instance = cls(**_constructor_kwargs(model_spec, args))
    if load_sd:
        instance.load_state_dict(model_spec['sd'])
    return instance
