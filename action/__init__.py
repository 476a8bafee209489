"""Import-path shim for the reference's ``action`` package."""
