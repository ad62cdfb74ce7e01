"""Trainer-level entry points mirroring the reference's ``implementations/<name>`` packages
(``main.py`` dispatches to ``implementations.<name>.main(parser)``, reference main.py:17-18)."""
