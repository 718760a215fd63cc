// RCCL-over-xGMI communication runtime: communicator + bucketed gradient reducer.
#pragma once
#include <pybind11/pybind11.h>
void bind_comm(pybind11::module_& m);
void bind_store(pybind11::module_& m);   // TcpStore rendezvous (store.cpp)
void bind_p2p(pybind11::module_& m);     // one-shot peer-memory all-reduce for small messages (p2p.hip)
