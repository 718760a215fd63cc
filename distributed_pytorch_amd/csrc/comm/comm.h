// RCCL-over-xGMI communication runtime: communicator + bucketed gradient reducer.
#pragma once
#include <pybind11/pybind11.h>
void bind_comm(pybind11::module_& m);
void bind_store(pybind11::module_& m);   // TcpStore rendezvous (store.cpp)
