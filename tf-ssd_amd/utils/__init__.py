"""Host-side mirrors of the reference's ``utils`` package (bbox / train / io / data / eval helpers) on top of libssd_hip."""
