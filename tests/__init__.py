"""Parity and host-logic tests of the MI355X engine (pytest markers: gpu / not gpu; see conftest.py)."""
