"""Derivative-spec bookkeeping shared by the interpolant classes (one module here; the reference keeps a
copy per class: barycentric.py:1173-1243, spline.py:447-517, slider.py:201-245).

Contract kept from the reference: ids are sequential from 0 in registration order and stable per orders
tuple; the messages of the ``ValueError`` / ``KeyError`` cases below.  Everything else -- one validation
helper that yields the canonical key, ``dict.setdefault`` as the registry, a two-flag dispatch for the
"orders xor id" rule -- is this package's own shape."""
from __future__ import annotations

import numpy as np


def canonical_orders(orders, ndim: int, max_order: int) -> tuple:
    """The orders as a tuple of Python ints, or ``ValueError`` naming the first offending entry."""
    if len(orders) != ndim:
        raise ValueError(f"derivative_order length {len(orders)} does not match num_dimensions {ndim}")

    def entry(axis, value):
        if not isinstance(value, (int, np.integer)):            # as in the reference, bool counts as int
            raise ValueError(f"derivative_order[{axis}] must be int, got {type(value).__name__}")
        if not 0 <= value <= max_order:
            raise ValueError(f"derivative_order[{axis}]={value} out of range [0, {max_order}]")
        return int(value)

    return tuple(entry(axis, value) for axis, value in enumerate(orders))


class DerivativeIdMixin:
    """Needs ``num_dimensions``, ``max_derivative_order``, ``_derivative_id_registry`` (dict: orders ->
    id) and ``_derivative_id_to_orders`` (list: id -> orders) on the host class."""

    def get_derivative_id(self, derivative_order) -> int:
        """Register a per-dimension derivative-orders tuple; the same tuple always maps to the same id."""
        orders = canonical_orders(derivative_order, self.num_dimensions, self.max_derivative_order)
        table = self._derivative_id_to_orders
        ident = self._derivative_id_registry.setdefault(orders, len(table))
        if ident == len(table):                   # first sighting: the id just handed out is the next slot
            table.append(orders)
        return ident

    def _resolve_derivative_args(self, derivative_order, derivative_id):
        """Exactly one of orders / id: ``ValueError`` for both or neither, ``KeyError`` for an id that
        was never registered.  Orders pass through untouched (range-checked where they are used)."""
        given = (derivative_order is not None, derivative_id is not None)
        if given == (True, True):
            raise ValueError("provide exactly one of derivative_order or derivative_id, not both")
        if given == (False, False):
            raise ValueError("must provide derivative_order or derivative_id")
        if given[0]:
            return derivative_order
        table = self._derivative_id_to_orders
        if not 0 <= derivative_id < len(table):
            raise KeyError(f"unknown derivative_id {derivative_id}; register via get_derivative_id() first")
        return list(table[derivative_id])
