"""RFI flagging algorithms (reference: src/katsdpsigproc/rfi/__init__.py)."""

#: ratio of the standard deviation to the median absolute deviation of a normal
#: distribution (reference rfi/__init__.py:31)
MAD_NORMAL = 1.4826
