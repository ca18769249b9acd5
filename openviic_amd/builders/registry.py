"""Name -> class registry behind the ``build_*`` factories.

Contract kept from the reference (``builders/registry.py:50-74``): ``register()`` works as a bare
call or as a decorator and keys the object by ``__name__``; registering a name twice is an
``AssertionError``; ``get`` of an unknown name is a ``KeyError``.
"""
from typing import Any, Callable, Dict, Iterator, Optional, Tuple


class Registry:
    def __init__(self, name: str) -> None:
        self._name = name
        self._table: Dict[str, Any] = {}

    @property
    def name(self) -> str:
        return self._name

    def _add(self, obj: Any) -> Any:
        key = obj.__name__
        assert key not in self._table, \
            "An object named '{}' was already registered in '{}' registry!".format(key, self._name)
        self._table[key] = obj
        return obj

    def register(self, obj: Optional[Any] = None) -> Any:
        if obj is None:
            return self._add          # @REGISTRY.register()
        self._add(obj)                # REGISTRY.register(cls)
        return None

    def get(self, name: str) -> Any:
        try:
            return self._table[name]
        except KeyError:
            raise KeyError("No object named '{}' found in '{}' registry!".format(name, self._name)) from None

    def names(self):
        return sorted(self._table)

    def __contains__(self, name: str) -> bool:
        return name in self._table

    def __iter__(self) -> Iterator[Tuple[str, Any]]:
        return iter(self._table.items())

    def __len__(self) -> int:
        return len(self._table)

    def __repr__(self) -> str:
        rows = "\n".join("  {:<48s} {}".format(k, v) for k, v in sorted(self._table.items()))
        return "Registry of {}:\n{}".format(self._name, rows)
